// ptz_view_kernels.h -- the packed problem of a bundle adjustment over a candidate subset of a RESIDENT rig, built on the device
// (ptz_ba_batch_create_views; included by ptz_ba.hip only).
//
// What the host does for ptz_ba_batch_create -- PTZRayOptimizer::Pack's walk over the tracks (AddConstraints2d2d's residual
// order, src/core/ptzray_optimizer.cc:799-850) and build_pairs' internal ray order and camera-major lists -- from the rig's
// tracks in HBM and the view's image -> compact-camera map.  Every array comes out word for word as the host builder's:
//   rays      = tracks with at least one candidate view, in track order (the caller's ray numbering), renumbered inside the
//               library by (candidate views descending, first candidate camera ascending, track ascending)
//   observations in (internal ray, image) order; weights = FULL track length (:805)
//   camera-major lists in ascending observation order
// The camera pairs, entry lists and runs then come from k_pairs as for any batch.
#ifndef PTZ_VIEW_KERNELS_H
#define PTZ_VIEW_KERNELS_H
#include <rocprim/block/block_radix_sort.hpp>
#include "ptz_ba_kernels.h"

namespace ptz {
namespace {

struct ViewDev {
  const int* trk_ptr;     // [n_track + 1] the rig's tracks
  const int* trk_img;     // [n_view] image ids, ascending inside a track
  const float2* trk_uv;   // [n_view] pixels
  int n_track, n_img, n_cam;
  int map_off;            // this view's image -> compact camera map starts here in cam_of_image
  int trk_off;            // first slot of this view's tracks in the batch-wide per-track arrays
  int n_ray, n_obs, max_len;  // filled by k_view_scan
};

struct ViewBuild {  // batch-wide arrays of the build (device)
  ViewDev* views;
  const int* cam_of_image;   // image -> compact camera of the view, or -1
  const int* cam_image;      // [total_cam] compact camera -> image
  int* t_len;                // [sum n_track] candidate views of a track
  int* t_first;              // ... its first candidate camera
  int* t_ext;                // ... its number among the tracks with a candidate view (the caller's ray index)
  unsigned long long* key_in;  // [sum n_track] sort keys: (internal-order key << 24 | track), all ones behind the rays
  int* val_in;
  int* ray_trk;              // [total_ray] track of internal ray j
  int* ray_len;              // [total_ray] its candidate views
  int* cam_cnt;              // [total_cam] observations of a camera
  int* chunk_cnt;            // per view [chunks of 256 internal rays][n_cam]: observations of camera c in the chunk, then (k_view_camscan)
                             // the number of its observations in the chunks in front
  const int* chunk_off;      // [n_view] where a view's block of chunk_cnt starts
  SceneDev* scene;
  // outputs (the batch's structure arrays)
  float2* obs_uv; int* obs_cam; int* obs_ray; int* ray_ptr; int* cam_ptr; int* cam_obs; int* wpos; int* cam_ray; float2* cam_uv;
  double* ray_w; int* ray_perm;
};

// thread = track: candidate views and first candidate camera
__global__ __launch_bounds__(256) void k_view_tracks(ViewBuild vb)
{
  const ViewDev v = vb.views[blockIdx.y];
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= v.n_track) return;
  const int* map = vb.cam_of_image + v.map_off;
  int len = 0, first = -1;
  for (int e = v.trk_ptr[t]; e < v.trk_ptr[t + 1]; ++e) {
    const int img = v.trk_img[e];
    const int c = (img >= 0 && img < v.n_img) ? map[img] : -1;
    if (c >= 0) { if (first < 0) first = c; ++len; }
  }
  vb.t_len[v.trk_off + t] = len;
  vb.t_first[v.trk_off + t] = first;
}

// exclusive block scan of one int per thread (1024 threads), running base carried by the caller; returns the exclusive prefix and
// adds the block's total to `base` (all threads)
__device__ __forceinline__ int block_excl_scan_1024(int x, int* wsum /* [17] */, int& base)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = x;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(incl, off, 64); if (lane >= off) incl += y; }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int woff = 0;
  for (int k = 0; k < w; ++k) woff += wsum[k];
  int total = 0;
  for (int k = 0; k < 16; ++k) total += wsum[k];
  const int excl = base + woff + incl - x;
  __syncthreads();
  base += total;
  return excl;
}

// one workgroup per view: which tracks are rays (t_ext = rank among them), how many rays / observations, the longest
__global__ __launch_bounds__(1024) void k_view_scan(ViewBuild vb)
{
  __shared__ int wsum[17];
  __shared__ int red[2][16];
  ViewDev& v = vb.views[blockIdx.x];
  int base = 0, nobs = 0, mlen = 0;
  for (int t0 = 0; t0 < v.n_track; t0 += 1024) {
    const int t = t0 + threadIdx.x;
    const int len = t < v.n_track ? vb.t_len[v.trk_off + t] : 0;
    const int ex = block_excl_scan_1024(len > 0 ? 1 : 0, wsum, base);
    if (t < v.n_track) vb.t_ext[v.trk_off + t] = ex;
    nobs += len;
    mlen = max(mlen, len);
  }
  // totals of the per-thread sums
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { nobs += __shfl_xor(nobs, off, 64); mlen = max(mlen, __shfl_xor(mlen, off, 64)); }
  if (lane == 0) { red[0][w] = nobs; red[1][w] = mlen; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int so = 0, sm = 0;
    for (int k = 0; k < 16; ++k) { so += red[0][k]; sm = max(sm, red[1][k]); }
    v.n_ray = base; v.n_obs = so; v.max_len = sm;
  }
}

// thread = track: the sort key of build_pairs' internal ray order -- (longest first, then first camera), ties in track order
constexpr unsigned long long VIEW_KEY_INVALID = 1ull << 46;
__global__ __launch_bounds__(256) void k_view_keys(ViewBuild vb)
{
  const ViewDev v = vb.views[blockIdx.y];
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= v.n_track) return;
  const int len = vb.t_len[v.trk_off + t];
  unsigned long long key = VIEW_KEY_INVALID | (unsigned long long)t;
  if (len > 0) key = ((unsigned long long)((v.max_len - len) * v.n_cam + vb.t_first[v.trk_off + t]) << 24) | (unsigned long long)t;
  key |= (unsigned long long)blockIdx.y << 47;  // views one after the other in the batch-wide sort
  vb.key_in[v.trk_off + t] = key;
  vb.val_in[v.trk_off + t] = t;
}

// One workgroup per view: keys and sort in ONE launch (views of up to 1024 * IPT tracks; the batch-wide rocprim::radix_sort_pairs over
// k_view_keys' 64-bit keys is 22 launches of a few microseconds each, a third of a view batch's build).  The same order: the key is
// (longest - length) * cameras + first camera, tracks without a candidate observation behind all others, ties in track order --
// rocprim::block_radix_sort is stable over the blocked arrangement (item = thread * IPT + i = track number).  Only the bits the
// view's largest key has are sorted.
template <int IPT>
__global__ __launch_bounds__(1024) void k_view_sort(ViewBuild vb, int* __restrict__ val_out)
{
  using Sort = rocprim::block_radix_sort<unsigned, 1024, IPT, int>;
  extern __shared__ __attribute__((aligned(16))) char view_sort_smem[];
  typename Sort::storage_type& st = *reinterpret_cast<typename Sort::storage_type*>(view_sort_smem);
  const ViewDev v = vb.views[blockIdx.x];
  const unsigned top = (unsigned)v.max_len * (unsigned)v.n_cam + (unsigned)v.n_cam;  // above every valid key (guarded < 2^22 by the host)
  int nb = 1;
  while ((1u << nb) <= top) ++nb;
  unsigned k[IPT];
  int t_of[IPT];
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int t = threadIdx.x * IPT + i;
    t_of[i] = t;
    unsigned key = 2u << nb;  // slots past the view's tracks: behind everything
    if (t < v.n_track) {
      const int len = vb.t_len[v.trk_off + t];
      key = len > 0 ? (unsigned)((v.max_len - len) * v.n_cam + vb.t_first[v.trk_off + t]) : (1u << nb);
    }
    k[i] = key;
  }
  Sort().sort(k, t_of, st, 0, nb + 2);
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int r = threadIdx.x * IPT + i;
    if (r < v.n_track) val_out[v.trk_off + r] = t_of[i];
  }
}

// thread = internal ray: its track, candidate views, weight, the caller's number
__global__ __launch_bounds__(256) void k_view_rays(ViewBuild vb, const int* __restrict__ val_sorted)
{
  const ViewDev v = vb.views[blockIdx.y];
  const SceneDev s = vb.scene[blockIdx.y];
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= v.n_ray) return;
  const int t = val_sorted[v.trk_off + j];
  vb.ray_trk[s.ray_off + j] = t;
  vb.ray_len[s.ray_off + j] = vb.t_len[v.trk_off + t];
  vb.ray_w[s.ray_off + j] = (double)(v.trk_ptr[t + 1] - v.trk_ptr[t]);  // FULL track length (ptzray_optimizer.cc:805)
  vb.ray_perm[s.ray_off + j] = vb.t_ext[v.trk_off + t];
}

// one workgroup per view: observation ranges of the internal rays; the scene's counts
__global__ __launch_bounds__(1024) void k_view_rayscan(ViewBuild vb, int ray_block)
{
  __shared__ int wsum[17];
  const ViewDev v = vb.views[blockIdx.x];
  SceneDev& s = vb.scene[blockIdx.x];
  int* rp = vb.ray_ptr + s.ray_off + s.idx;
  int base = 0;
  for (int j0 = 0; j0 < v.n_ray; j0 += 1024) {
    const int j = j0 + threadIdx.x;
    const int len = j < v.n_ray ? vb.ray_len[s.ray_off + j] : 0;
    const int ex = block_excl_scan_1024(len, wsum, base);
    if (j < v.n_ray) rp[j] = s.obs_off + ex;
  }
  if (threadIdx.x == 0) {
    rp[v.n_ray] = s.obs_off + base;
    s.n_ray = v.n_ray;
    s.n_obs = v.n_obs;
    s.n_wave = (v.n_ray + 63) / 64;
    s.n_chunk = (v.n_ray + ray_block - 1) / ray_block;
  }
  for (int c = threadIdx.x; c < v.n_cam; c += 1024) vb.cam_cnt[s.cam_off + c] = 0;
}

// thread = internal ray: its observation records, (track, image) order; observations per camera
__global__ __launch_bounds__(256) void k_view_obs(ViewBuild vb)
{
  const ViewDev v = vb.views[blockIdx.y];
  const SceneDev s = vb.scene[blockIdx.y];
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= v.n_ray) return;
  const int t = vb.ray_trk[s.ray_off + j];
  const int* map = vb.cam_of_image + v.map_off;
  int a = vb.ray_ptr[s.ray_off + s.idx + j];
  for (int e = v.trk_ptr[t]; e < v.trk_ptr[t + 1]; ++e) {
    const int img = v.trk_img[e];
    const int c = (img >= 0 && img < v.n_img) ? map[img] : -1;
    if (c < 0) continue;
    vb.obs_uv[a] = v.trk_uv[e];
    vb.obs_cam[a] = c;
    vb.obs_ray[a] = j;
    atomicAdd(&vb.cam_cnt[s.cam_off + c], 1);  // (integer: the total does not depend on the order)
    ++a;
  }
}

// ---- camera-major lists by counting (views of up to VIEW_LDS_CAMS cameras; wider ones take k_view_obs / k_view_camlists_wide) --------
// The camera-major lists hold a camera's observations in ascending observation order, i.e. ascending internal ray: a stable
// counting sort of the (ray, image)-ordered observations by camera.  Three launches: (1) per chunk of 256 rays, how many
// observations every camera has in it (LDS histogram; integer sums: no order dependence); (2) per camera, the running sum
// over the chunks (k_view_camscan); (3) per chunk again, an observation's slot = the camera's range + the chunks in front + its rank
// inside the chunk -- the lanes in front of it in its wave that see the same camera (one 64-bit lane mask per wave and camera,
// set with LDS atomics -- an OR, so no order dependence either) + the counts of the waves in front.
constexpr int VIEW_LDS_CAMS = 2048;
__global__ __launch_bounds__(256) void k_view_hist(ViewBuild vb)
{
  extern __shared__ int vh_lds[];
  const ViewDev v = vb.views[blockIdx.y];
  if ((int)blockIdx.x * 256 >= v.n_ray) return;
  const SceneDev s = vb.scene[blockIdx.y];
  for (int c = threadIdx.x; c < v.n_cam; c += 256) vh_lds[c] = 0;
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < v.n_ray) {
    const int t = vb.ray_trk[s.ray_off + j];
    const int* map = vb.cam_of_image + v.map_off;
    for (int e = v.trk_ptr[t]; e < v.trk_ptr[t + 1]; ++e) {
      const int img = v.trk_img[e];
      const int c = (img >= 0 && img < v.n_img) ? map[img] : -1;
      if (c >= 0) atomicAdd(&vh_lds[c], 1);
    }
  }
  __syncthreads();
  int* out = vb.chunk_cnt + vb.chunk_off[blockIdx.y] + (size_t)blockIdx.x * v.n_cam;
  for (int c = threadIdx.x; c < v.n_cam; c += 256) out[c] = vh_lds[c];
}

// one workgroup per view: per camera the running sum over the chunks (left in chunk_cnt), then the cameras' observation ranges
__global__ __launch_bounds__(1024) void k_view_chunkscan(ViewBuild vb)
{
  __shared__ int wsum[17];
  const ViewDev v = vb.views[blockIdx.x];
  const SceneDev s = vb.scene[blockIdx.x];
  int* cp = vb.cam_ptr + s.cam_off + s.idx;
  int* cc = vb.chunk_cnt + vb.chunk_off[blockIdx.x];
  const int nch = (v.n_ray + 255) / 256;
  int base = 0;
  for (int c0 = 0; c0 < v.n_cam; c0 += 1024) {
    const int c = c0 + threadIdx.x;
    int run = 0;
    if (c < v.n_cam) {
      int k = 0;
      for (; k + 8 <= nch; k += 8) {  // eight loads in flight
        int x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = cc[(size_t)(k + u) * v.n_cam + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) { cc[(size_t)(k + u) * v.n_cam + c] = run; run += x[u]; }
      }
      for (; k < nch; ++k) { const int x = cc[(size_t)k * v.n_cam + c]; cc[(size_t)k * v.n_cam + c] = run; run += x; }
    }
    const int ex = block_excl_scan_1024(run, wsum, base);
    if (c < v.n_cam) cp[c] = s.obs_off + ex;
  }
  if (threadIdx.x == 0) cp[v.n_cam] = s.obs_off + base;
}

// thread = internal ray: its observation records in (track, image) order and their places in the camera-major lists
__global__ __launch_bounds__(256) void k_view_place(ViewBuild vb)
{
  extern __shared__ unsigned long long vp_mask[];  // [4][n_cam]
  const ViewDev v = vb.views[blockIdx.y];
  if ((int)blockIdx.x * 256 >= v.n_ray) return;
  const SceneDev s = vb.scene[blockIdx.y];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < 4 * v.n_cam; c += 256) vp_mask[c] = 0ull;
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int* map = vb.cam_of_image + v.map_off;
  int e0 = 0, e1 = 0;
  if (j < v.n_ray) {
    const int t = vb.ray_trk[s.ray_off + j];
    e0 = v.trk_ptr[t]; e1 = v.trk_ptr[t + 1];
    for (int e = e0; e < e1; ++e) {
      const int img = v.trk_img[e];
      const int c = (img >= 0 && img < v.n_img) ? map[img] : -1;
      if (c >= 0) atomicOr(&vp_mask[(size_t)w * v.n_cam + c], 1ull << lane);
    }
  }
  __syncthreads();
  if (j >= v.n_ray) return;
  const int* cp = vb.cam_ptr + s.cam_off + s.idx;
  const int* cb = vb.chunk_cnt + vb.chunk_off[blockIdx.y] + (size_t)blockIdx.x * v.n_cam;
  int a = vb.ray_ptr[s.ray_off + s.idx + j];
  for (int e = e0; e < e1; ++e) {
    const int img = v.trk_img[e];
    const int c = (img >= 0 && img < v.n_img) ? map[img] : -1;
    if (c < 0) continue;
    int rank = __popcll(vp_mask[(size_t)w * v.n_cam + c] & ((1ull << lane) - 1ull));
    for (int k = 0; k < w; ++k) rank += __popcll(vp_mask[(size_t)k * v.n_cam + c]);
    const int slot = cp[c] + cb[c] + rank;
    const float2 uv = v.trk_uv[e];
    vb.obs_uv[a] = uv; vb.obs_cam[a] = c; vb.obs_ray[a] = j;
    vb.cam_obs[slot] = a; vb.cam_ray[slot] = s.ray_off + j; vb.cam_uv[slot] = uv; vb.wpos[a] = slot;
    ++a;
  }
}


// one workgroup per view: observation ranges of the cameras
__global__ __launch_bounds__(1024) void k_view_camscan(ViewBuild vb)
{
  __shared__ int wsum[17];
  const ViewDev v = vb.views[blockIdx.x];
  const SceneDev s = vb.scene[blockIdx.x];
  int* cp = vb.cam_ptr + s.cam_off + s.idx;
  int base = 0;
  for (int c0 = 0; c0 < v.n_cam; c0 += 1024) {
    const int c = c0 + threadIdx.x;
    const int n = c < v.n_cam ? vb.cam_cnt[s.cam_off + c] : 0;
    const int ex = block_excl_scan_1024(n, wsum, base);
    if (c < v.n_cam) cp[c] = s.obs_off + ex;
  }
  if (threadIdx.x == 0) cp[v.n_cam] = s.obs_off + base;
}

// (views wider than VIEW_LDS_CAMS) one workgroup per (view, camera): the camera's observations in ascending observation order = ascending internal ray.
// Every thread asks whether its ray's track has the camera's image (binary search: images ascend inside a track); the hits
// of a sweep of 256 rays are ranked by ballot, waves in wave order.
__global__ __launch_bounds__(256) void k_view_camlists_wide(ViewBuild vb)
{
  __shared__ int wcnt[4];
  __shared__ int out_s;
  const ViewDev v = vb.views[blockIdx.y];
  const SceneDev s = vb.scene[blockIdx.y];
  const int c = blockIdx.x;
  if (c >= v.n_cam) return;
  const int img_c = vb.cam_image[s.cam_off + c];
  const int* map = vb.cam_of_image + v.map_off;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) out_s = vb.cam_ptr[s.cam_off + s.idx + c];
  __syncthreads();
  const int out_end = vb.cam_ptr[s.cam_off + s.idx + c + 1];
  for (int j0 = 0; j0 < v.n_ray; j0 += 256) {
    if (out_s >= out_end) break;  // (uniform: every observation of the camera is placed)
    const int j = j0 + threadIdx.x;
    int a = -1;
    float2 uv = make_float2(0.f, 0.f);
    if (j < v.n_ray) {
      const int t = vb.ray_trk[s.ray_off + j];
      int lo = v.trk_ptr[t], hi = v.trk_ptr[t + 1];
      const int e0 = lo;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (v.trk_img[mid] < img_c) lo = mid + 1; else hi = mid; }
      if (lo < v.trk_ptr[t + 1] && v.trk_img[lo] == img_c) {
        int rank = 0;  // candidate views of the track in front of this one
        for (int e = e0; e < lo; ++e) { const int im = v.trk_img[e]; rank += (im >= 0 && im < v.n_img && map[im] >= 0) ? 1 : 0; }
        a = vb.ray_ptr[s.ray_off + s.idx + j] + rank;
        uv = v.trk_uv[lo];
      }
    }
    const unsigned long long m = __ballot(a >= 0);
    if (lane == 0) wcnt[w] = __popcll(m);
    __syncthreads();
    int off = out_s;
    for (int k = 0; k < w; ++k) off += wcnt[k];
    if (a >= 0) {
      const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
      vb.cam_obs[slot] = a;
      vb.cam_ray[slot] = s.ray_off + j;
      vb.cam_uv[slot] = uv;
      vb.wpos[a] = slot;
    }
    __syncthreads();
    if (threadIdx.x == 0) out_s += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
}

// after k_pair_scan: the scenes' pair counts
__global__ void k_view_pairs_patch(SceneDev* scene, int n, const int* __restrict__ scene_tot)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) scene[i].n_pair = scene_tot[6 * i];
}

// one workgroup per view: tile-level adjacency of the reduced camera system (what ptz_ba_batch_create marks from the pair list:
// a camera's own tiles, the tiles of every camera pair, the dense tail rows), natural tile order, lower triangle
__global__ __launch_bounds__(256) void k_view_adjacency(const SceneDev* __restrict__ scene, const int* __restrict__ pair_ci,
                                                        const int* __restrict__ pair_cj, int NC, int nt, unsigned char* __restrict__ m0)
{
  const SceneDev s = scene[blockIdx.x];
  unsigned char* m = m0 + (size_t)blockIdx.x * nt * nt;
  for (int i = threadIdx.x; i < nt * nt; i += 256) m[i] = 0;
  __syncthreads();
  auto tile_lo = [&](int cam) { return (cam * NC) / CHOL_NB; };
  auto tile_hi = [&](int cam) { return (cam * NC + NC - 1) / CHOL_NB; };
  for (int c = threadIdx.x; c < s.n_cam; c += 256)
    for (int a = tile_lo(c); a <= tile_hi(c); ++a)
      for (int e = tile_lo(c); e <= a; ++e) m[a * nt + e] = 1;
  for (int p = threadIdx.x; p < s.n_pair; p += 256) {
    const int ci = pair_ci[s.pair_off + p], cj = pair_cj[s.pair_off + p];
    for (int a = tile_lo(ci); a <= tile_hi(ci); ++a)
      for (int e = tile_lo(cj); e <= tile_hi(cj); ++e) {
        if (a >= e) m[a * nt + e] = 1; else m[e * nt + a] = 1;
      }
  }
  const int first_dense = (s.n_cam * NC) / CHOL_NB;
  for (int i = threadIdx.x; i < nt * nt; i += 256) {
    const int a = i / nt, e = i % nt;
    if (a >= first_dense && e <= a) m[i] = 1;
  }
}

// Pix2Ray on the device with the reference's operation order (ptzray_optimizer.cc:768-797 as PTZRayOptimizer::Pack evaluates it on
// the host): t = RKinv [u, v, 1] (row by row, left to right), n = |t|, acc += t / n over the track's candidate views in image
// order, acc /= count, ray = acc / |acc|.  No contraction: every product and sum rounds as the host's.
__global__ __launch_bounds__(256) void k_view_pix2ray(Dev d, const double* __restrict__ cam0, const double* __restrict__ rkinv, double* __restrict__ ray0)
{
#pragma clang fp contract(off)  // (HIP's __dmul_rn / __dadd_rn are plain operators: without this the compiler fuses them)
  const int sc = blockIdx.y;
  const SceneDev s = d.scene[sc];
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= s.n_ray) return;
  (void)cam0;
  const int* rp = d.ray_ptr + s.ray_off + s.idx;
  double a0 = 0, a1 = 0, a2 = 0;
  int cnt = 0;
  for (int a = rp[j]; a < rp[j + 1]; ++a) {
    const double* M = rkinv + (size_t)(s.cam_off + d.obs_cam[a]) * 9;
    const float2 uv = d.obs_uv[a];
    const double x = (double)uv.x, y = (double)uv.y;
    const double t0 = (M[0] * x + M[1] * y) + M[2];
    const double t1 = (M[3] * x + M[4] * y) + M[5];
    const double t2 = (M[6] * x + M[7] * y) + M[8];
    const double n = sqrt((t0 * t0 + t1 * t1) + t2 * t2);
    a0 = a0 + t0 / n; a1 = a1 + t1 / n; a2 = a2 + t2 / n;
    ++cnt;
  }
  const double dc = (double)cnt;
  a0 = a0 / dc; a1 = a1 / dc; a2 = a2 / dc;
  const double n = sqrt((a0 * a0 + a1 * a1) + a2 * a2);
  double* out = ray0 + (size_t)(s.ray_off + j) * 3;
  out[0] = a0 / n; out[1] = a1 / n; out[2] = a2 / n;
}

}  // namespace
}  // namespace ptz
#endif  // PTZ_VIEW_KERNELS_H
