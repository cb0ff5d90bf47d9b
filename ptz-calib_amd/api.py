"""ctypes binding of include/ptz_calib_amd.h (libptzcalib_hip.so).

No compute happens in Python and there is no CPU fallback: if the HIP library is missing or no device is
present, every compute call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTZCALIB_LIB", os.path.join(_HERE, "libptzcalib_hip.so"))  # override: A/B probe builds only

CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2
BA_PTZRay, BA_PTZRayDist, BA_PTZRayFxfyDist, BA_PTZRayDistDisp = 0, 1, 2, 3
KRT_F, KRT_FDist, KRT_Fxfy, KRT_FxfyDist = 0, 1, 2, 3
PROF_SLOTS = 16
_ERR = {-1: "PTZ_EINVAL", -2: "PTZ_ENODEVICE", -3: "PTZ_ENOMEM", -4: "PTZ_EUNSUPPORTED", -5: "PTZ_ELIMIT", -6: "PTZ_ENOOBS"}

EXPORTS = ["ptz_lm_options_default", "ptz_version", "ptz_device_count", "ptz_ba_batch_create", "ptz_ba_batch_destroy",
           "ptz_ba_batch_set_state", "ptz_ba_batch_solve", "ptz_ba_batch_get_state", "ptz_ba_batch_last_solve_ms",
           "ptz_ba_batch_set_profiling", "ptz_ba_batch_get_profile", "ptz_ba_solve", "ptz_ba_cam_block_dim",
           "ptz_ba_batch_linearize", "ptz_ba_batch_pix2ray", "ptz_ba_batch_cam_block_dim", "ptz_chol_solve_batch", "ptz_krt_solve_batch",
           "ptz_krt_solve_batch_2d3d", "ptz_krt_solve_batch_device", "ptz_trim_cache", "ptz_mfma_f64_peak", "ptz_ba_solve_sharded",
           "ptz_krt_solve_batch_sharded", "ptz_hbm_bandwidth", "ptz_ba_batch_set_disp", "ptz_ba_batch_get_disp", "ptz_ba_solve_disp",
           "ptz_ba_plan_tile_order", "ptz_rig_create", "ptz_rig_destroy", "ptz_ba_batch_create_views", "ptz_ba_batch_set_state_pix2ray",
           "ptz_debug_batch_structure_hash", "ptz_debug_batch_initial_rays", "ptz_krt_table_create", "ptz_krt_table_destroy",
           "ptz_krt_solve_attempts"]


class PtzError(RuntimeError):
    def __init__(self, code, where):
        super().__init__(f"{where}: {_ERR.get(code, code)}")
        self.code = code


class LmOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("device_id", C.c_int32),
                ("max_num_consecutive_invalid_steps", C.c_int32), ("jacobi_scaling", C.c_int32),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double), ("function_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("krt_lanes_per_query", C.c_int32), ("reserved_", C.c_int32)]


class LmSummary(C.Structure):
    _fields_ = [("termination_type", C.c_int32), ("num_iterations", C.c_int32), ("num_lm_steps", C.c_int32),
                ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
                ("num_residuals", C.c_int32), ("num_linear_solves", C.c_int32), ("num_jacobian_evals", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("final_radius", C.c_double),
                ("final_gradient_max_norm", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class BaProblem(C.Structure):
    _fields_ = [("n_cam", C.c_int32), ("n_ray", C.c_int32), ("n_obs", C.c_int64), ("obs_uv", C.c_void_p),
                ("obs_cam", C.c_void_p), ("obs_ray", C.c_void_p), ("ray_weight", C.c_void_p),
                ("n_obs3d", C.c_int32), ("obs3d_uv", C.c_void_p), ("obs3d_xyz", C.c_void_p),
                ("obs3d_cam", C.c_void_p), ("factor_type", C.c_int32), ("ic_of_cam", C.c_void_p)]


_lib = None


def lib():
    """Load the HIP library; fails loudly if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
        _lib = C.CDLL(LIB_PATH)
        _lib.ptz_version.restype = C.c_char_p
        _lib.ptz_device_count.restype = C.c_int32
        _lib.ptz_ba_cam_block_dim.restype = C.c_int32
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _check(rc, where):
    if rc != 0:
        raise PtzError(rc, where)


def version() -> str:
    return lib().ptz_version().decode()


def trim_cache() -> None:
    """Give the library's parked device blocks / streams / events back to the driver (ptz_trim_cache)."""
    lib().ptz_trim_cache()


def device_count() -> int:
    return int(lib().ptz_device_count())


def default_options(**kw) -> LmOptions:
    o = LmOptions()
    lib().ptz_lm_options_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _pack_problem(sc, keep):
    arrs = dict(uv=np.ascontiguousarray(sc.obs_uv, dtype=np.float32), cam=np.ascontiguousarray(sc.obs_cam, dtype=np.int32),
                ray=np.ascontiguousarray(sc.obs_ray, dtype=np.int32), w=np.ascontiguousarray(sc.ray_weight, dtype=np.float64))
    keep.append(arrs)
    p = BaProblem()
    p.n_cam, p.n_ray, p.n_obs = sc.n_cam, sc.n_ray, len(arrs["cam"])
    p.obs_uv, p.obs_cam, p.obs_ray, p.ray_weight = _p(arrs["uv"]), _p(arrs["cam"]), _p(arrs["ray"]), _p(arrs["w"])
    o3 = getattr(sc, "obs3d", None)
    if o3 is not None and len(o3["cam"]) > 0:
        arrs["o3uv"] = np.ascontiguousarray(o3["uv"], dtype=np.float32)
        arrs["o3xyz"] = np.ascontiguousarray(o3["xyz"], dtype=np.float64)
        arrs["o3cam"] = np.ascontiguousarray(o3["cam"], dtype=np.int32)
        p.n_obs3d = len(arrs["o3cam"])
        p.obs3d_uv, p.obs3d_xyz, p.obs3d_cam = _p(arrs["o3uv"]), _p(arrs["o3xyz"]), _p(arrs["o3cam"])
    else:
        p.n_obs3d = 0
    p.factor_type = sc.factor_type
    ic = getattr(sc, "ic_of_cam", None)
    if ic is not None:
        arrs["ic"] = np.ascontiguousarray(ic, dtype=np.int32)
        if len(arrs["ic"]) != sc.n_cam:
            raise ValueError("ic_of_cam must have one id per camera")
        p.ic_of_cam = _p(arrs["ic"])
    return p


class BaBatch:
    """Device-resident batch of independent PTZ-IBA problems (ptz_ba_batch_*)."""

    def __init__(self, scenes, **opt):
        self.scenes = list(scenes)
        self.n = len(self.scenes)
        keep = []
        probs = (BaProblem * self.n)(*[_pack_problem(s, keep) for s in self.scenes])
        self.opt = default_options(**opt)
        self.handle = C.c_void_p()
        _check(lib().ptz_ba_batch_create(self.n, probs, C.byref(self.opt), C.byref(self.handle)), "ptz_ba_batch_create")
        self.cam_off = np.concatenate([[0], np.cumsum([s.n_cam for s in self.scenes])])
        self.ray_off = np.concatenate([[0], np.cumsum([s.n_ray for s in self.scenes])])
        self.obs_off = np.concatenate([[0], np.cumsum([s.n_obs for s in self.scenes])])
        self.nw = int(lib().ptz_ba_cam_block_dim(self.scenes[0].factor_type))
        self.nc = int(lib().ptz_ba_batch_cam_block_dim(self.handle))

    def close(self):
        if self.handle:
            lib().ptz_ba_batch_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_state(self, cams=None, rays=None, tlws=None):
        cam = np.ascontiguousarray(np.concatenate([s.cam_init for s in self.scenes] if cams is None else cams), dtype=np.float64)
        ray = np.ascontiguousarray(np.concatenate([s.ray_init for s in self.scenes] if rays is None else rays), dtype=np.float64)
        if tlws is None and any(getattr(s, "tlw_init", None) is not None for s in self.scenes):
            tlws = [getattr(s, "tlw_init", None) if getattr(s, "tlw_init", None) is not None else np.zeros(6) for s in self.scenes]
        tlw = None if tlws is None else np.ascontiguousarray(np.stack(tlws), dtype=np.float64)
        _check(lib().ptz_ba_batch_set_state(self.handle, _p(cam), _p(ray), _p(tlw)), "ptz_ba_batch_set_state")

    def set_disp(self, disps):
        """PTZRayDistDisp: the initial displacement block (d0, d1, d2) of every scene ([n, 3]; zeros unless set)."""
        d = np.ascontiguousarray(np.asarray(disps, dtype=np.float64).reshape(self.n, 3))
        _check(lib().ptz_ba_batch_set_disp(self.handle, _p(d)), "ptz_ba_batch_set_disp")

    def get_disp(self):
        d = np.zeros((self.n, 3))
        _check(lib().ptz_ba_batch_get_disp(self.handle, _p(d)), "ptz_ba_batch_get_disp")
        return d

    def pix2ray(self):
        _check(lib().ptz_ba_batch_pix2ray(self.handle), "ptz_ba_batch_pix2ray")

    def solve(self):
        summ = (LmSummary * self.n)()
        _check(lib().ptz_ba_batch_solve(self.handle, summ), "ptz_ba_batch_solve")
        return [s.as_dict() for s in summ]

    def get_state(self):
        cam = np.zeros((int(self.cam_off[-1]), 15))
        ray = np.zeros((int(self.ray_off[-1]), 3))
        tlw = np.zeros((self.n, 6))
        _check(lib().ptz_ba_batch_get_state(self.handle, _p(cam), _p(ray), _p(tlw)), "ptz_ba_batch_get_state")
        cams = [cam[self.cam_off[i]:self.cam_off[i + 1]] for i in range(self.n)]
        rays = [ray[self.ray_off[i]:self.ray_off[i + 1]] for i in range(self.n)]
        self.last_tlw = tlw
        return cams, rays

    def last_solve_ms(self) -> float:
        ms = C.c_double()
        _check(lib().ptz_ba_batch_last_solve_ms(self.handle, C.byref(ms)), "ptz_ba_batch_last_solve_ms")
        return ms.value

    def set_profiling(self, enable: bool):
        _check(lib().ptz_ba_batch_set_profiling(self.handle, int(enable)), "ptz_ba_batch_set_profiling")

    def get_profile(self):
        ms = (C.c_double * PROF_SLOTS)()
        cnt = (C.c_int64 * PROF_SLOTS)()
        names = (C.c_char_p * PROF_SLOTS)()
        _check(lib().ptz_ba_batch_get_profile(self.handle, ms, cnt, names), "ptz_ba_batch_get_profile")
        return {names[i].decode(): {"ms": ms[i], "launches": cnt[i]} for i in range(PROF_SLOTS) if names[i]}

    def linearize(self, index=0):
        s = self.scenes[index]
        nc = self.nc
        cost = C.c_double()
        g_c = np.zeros((s.n_cam, nc)); U = np.zeros((s.n_cam, nc, nc))
        g_r = np.zeros((s.n_ray, 3)); V = np.zeros((s.n_ray, 3, 3)); W = np.zeros((s.n_obs, self.nw, 3))
        _check(lib().ptz_ba_batch_linearize(self.handle, index, C.byref(cost), _p(g_c), _p(U), _p(g_r), _p(V), _p(W)),
               "ptz_ba_batch_linearize")
        return dict(cost=cost.value, g_c=g_c, U=U, g_r=g_r, V=V, W=W, nc=nc)


class RigView(C.Structure):
    _fields_ = [("rig", C.c_void_p), ("n_cam", C.c_int32), ("cam_image", C.c_void_p)]


class Rig:
    """One rig's tracks resident in HBM (ptz_rig_create): trk_ptr [n_track + 1], per view the image id (ascending inside a
    track) and the pixel."""

    def __init__(self, n_img, trk_ptr, trk_img, trk_uv, device_id=0):
        self.n_img = int(n_img)
        self.trk_ptr = np.ascontiguousarray(trk_ptr, dtype=np.int64)
        self.trk_img = np.ascontiguousarray(trk_img, dtype=np.int32)
        self.trk_uv = np.ascontiguousarray(trk_uv, dtype=np.float32)
        self.handle = C.c_void_p()
        _check(lib().ptz_rig_create(self.n_img, len(self.trk_ptr) - 1, _p(self.trk_ptr), _p(self.trk_img), _p(self.trk_uv), int(device_id),
                                    C.byref(self.handle)), "ptz_rig_create")

    @classmethod
    def from_scene(cls, sc, device_id=0):
        """The tracks of a synthetic scene (its observations are (track, image)-ordered: every ray is one track)."""
        ptr = np.concatenate([[0], np.cumsum(np.bincount(sc.obs_ray, minlength=sc.n_ray))]).astype(np.int64)
        return cls(sc.n_cam, ptr, sc.obs_cam, sc.obs_uv, device_id)

    def close(self):
        if self.handle:
            lib().ptz_rig_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def view_problem(sc, images):
    """The packed problem of the candidate set `images` (ascending) of a synthetic scene, as PTZRayOptimizer::Pack makes it on the
    host: observations of candidate views in (track, image) order, tracks without one dropped, cameras and rays renumbered,
    weights = FULL track length (ptzray_optimizer.cc:805).  Returns a scene-like object for BaBatch."""
    import copy
    images = np.asarray(images, dtype=np.int64)
    cmap = -np.ones(sc.n_cam, dtype=np.int64)
    cmap[images] = np.arange(len(images))
    keep = cmap[sc.obs_cam] >= 0
    v = copy.copy(sc)
    v.n_cam = len(images)
    v.obs_uv = sc.obs_uv[keep]
    v.obs_cam = cmap[sc.obs_cam[keep]].astype(np.int32)
    tracks, new_ray = np.unique(sc.obs_ray[keep], return_inverse=True)
    v.obs_ray = new_ray.astype(np.int32)
    v.n_ray = len(tracks)
    v.ray_weight = np.bincount(sc.obs_ray, minlength=sc.n_ray)[tracks].astype(np.float64)
    v.cam_init = sc.cam_init[images].copy(); v.cam_gt = sc.cam_gt[images].copy()
    v.ray_init = sc.ray_init[tracks].copy(); v.ray_gt = sc.ray_gt[tracks].copy()
    v.ic_of_cam = None
    v.view_tracks = tracks
    return v


class ViewBatch(BaBatch):
    """ptz_ba_batch_create_views: the batch of the packed problems of (rig, candidate images) views, built on the device."""

    def __init__(self, rigs, image_lists, factor_type=0, **opt):
        self.scenes = []
        self.n = len(rigs)
        self._keep = [np.ascontiguousarray(im, dtype=np.int32) for im in image_lists]
        views = (RigView * self.n)()
        for i, (rg, im) in enumerate(zip(rigs, self._keep)):
            views[i].rig = rg.handle
            views[i].n_cam = len(im)
            views[i].cam_image = _p(im)
        self.opt = default_options(**opt)
        self.handle = C.c_void_p()
        _check(lib().ptz_ba_batch_create_views(self.n, views, int(factor_type), C.byref(self.opt), C.byref(self.handle)), "ptz_ba_batch_create_views")
        self.nw = int(lib().ptz_ba_cam_block_dim(int(factor_type)))
        self.nc = int(lib().ptz_ba_batch_cam_block_dim(self.handle))
        self.n_cams = [len(im) for im in self._keep]

    def set_state_pix2ray(self, cams, rkinv):
        cam = np.ascontiguousarray(np.concatenate(cams), dtype=np.float64)
        rk = np.ascontiguousarray(np.concatenate(rkinv), dtype=np.float64)
        _check(lib().ptz_ba_batch_set_state_pix2ray(self.handle, _p(cam), _p(rk)), "ptz_ba_batch_set_state_pix2ray")

    def get_cams(self):
        cam = np.zeros((int(sum(self.n_cams)), 15))
        tlw = np.zeros((self.n, 6))
        _check(lib().ptz_ba_batch_get_state(self.handle, _p(cam), None, _p(tlw)), "ptz_ba_batch_get_state")
        off = np.concatenate([[0], np.cumsum(self.n_cams)])
        return [cam[off[i]:off[i + 1]] for i in range(self.n)]


def initial_rays(batch, total_rays) -> np.ndarray:
    r = np.zeros((int(total_rays), 3))
    _check(lib().ptz_debug_batch_initial_rays(batch.handle, _p(r)), "ptz_debug_batch_initial_rays")
    return r


def structure_hash(batch) -> int:
    h = C.c_uint64()
    _check(lib().ptz_debug_batch_structure_hash(batch.handle, C.byref(h)), "ptz_debug_batch_structure_hash")
    return int(h.value)


def ba_solve(scene, cam0=None, ray0=None, tlw0=None, return_tlw=False, **opt):
    """One-shot ptz_ba_solve.  Returns (cam, ray, summary dict) [+ tlw when return_tlw]."""
    keep = []
    p = _pack_problem(scene, keep)
    cam = np.array(scene.cam_init if cam0 is None else cam0, dtype=np.float64, order="C").copy()
    ray = np.array(scene.ray_init if ray0 is None else ray0, dtype=np.float64, order="C").copy()
    if tlw0 is None:
        tlw0 = getattr(scene, "tlw_init", None)
    tlw = np.zeros(6) if tlw0 is None else np.array(tlw0, dtype=np.float64).copy()
    o = default_options(**opt)
    s = LmSummary()
    _check(lib().ptz_ba_solve(C.byref(p), _p(cam), _p(ray), _p(tlw), C.byref(o), C.byref(s)), "ptz_ba_solve")
    return (cam, ray, s.as_dict(), tlw) if return_tlw else (cam, ray, s.as_dict())


def ba_solve_disp(scene, cam0=None, ray0=None, tlw0=None, disp0=None, **opt):
    """One-shot ptz_ba_solve_disp (PTZRayDistDisp).  Returns (cam, ray, summary dict, tlw, disp).

    The returned camera vectors hold the refined PARAMETER blocks: t_z (cam[:, 9]) comes back as it went in.  The reference's
    ObtainRefinedCameraParams additionally folds the refined displacement into it, t_z += d0 + d1 fx + d2 fx^2
    (ptzray_optimizer.cc:693, 714); the C++ class PTZRayOptimizer does that on read-back, this C-ABI level (and
    ptz_ba_batch_get_state / get_disp) hands out the two blocks separately -- `fold_displacement(cam, disp)` applies it."""
    keep = []
    p = _pack_problem(scene, keep)
    cam = np.array(scene.cam_init if cam0 is None else cam0, dtype=np.float64, order="C").copy()
    ray = np.array(scene.ray_init if ray0 is None else ray0, dtype=np.float64, order="C").copy()
    if tlw0 is None:
        tlw0 = getattr(scene, "tlw_init", None)
    tlw = np.zeros(6) if tlw0 is None else np.array(tlw0, dtype=np.float64).copy()
    disp = np.zeros(3) if disp0 is None else np.array(disp0, dtype=np.float64).copy()
    o = default_options(**opt)
    s = LmSummary()
    _check(lib().ptz_ba_solve_disp(C.byref(p), _p(cam), _p(ray), _p(tlw), _p(disp), C.byref(o), C.byref(s)), "ptz_ba_solve_disp")
    return cam, ray, s.as_dict(), tlw, disp


def fold_displacement(cam, disp):
    """ObtainRefinedCameraParams' last step for PTZRayDistDisp (ptzray_optimizer.cc:693, 714): t_z += d0 + d1 fx + d2 fx^2."""
    out = np.array(cam, dtype=np.float64).copy()
    out[:, 9] += disp[0] + disp[1] * out[:, 0] + disp[2] * out[:, 0] ** 2
    return out


def plan_tile_order(mask, first_dense):
    """Elimination order of a tile graph (host logic of ptz_ba_batch_create).  mask: [nt, nt] lower-triangular adjacency.
    Returns (planned, perm, (lane_a, lane_b))."""
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    nt = m.shape[0]
    perm = np.zeros(nt, dtype=np.int32)
    lanes = np.zeros(2, dtype=np.int32)
    sched = np.zeros((nt, 4), dtype=np.int32)
    n_steps = np.zeros(1, dtype=np.int32)
    rc = lib().ptz_ba_plan_tile_order(nt, int(first_dense), _p(m), _p(perm), _p(lanes), _p(sched), _p(n_steps))
    if rc < 0:
        raise PtzError(rc, "ptz_ba_plan_tile_order")
    plan_tile_order.last_schedule = sched[:int(n_steps[0])]
    return bool(rc), perm, (int(lanes[0]), int(lanes[1]))


def chol_solve_batch(A, rhs, device_id=0):
    """A: [count, n, n] SPD (lower triangle read), rhs: [count, n].  Returns (x, fail, device_ms)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    rhs = np.ascontiguousarray(rhs, dtype=np.float64)
    count, n = A.shape[0], A.shape[1]
    x = np.zeros((count, n))
    fail = np.zeros(count, dtype=np.int32)
    ms = C.c_double()
    _check(lib().ptz_chol_solve_batch(count, n, _p(A), _p(rhs), _p(x), _p(fail), device_id, C.byref(ms)), "ptz_chol_solve_batch")
    return x, fail, ms.value


def krt_solve_batch(batch, max_reproj_error=100.0, **opt):
    """batch: synth.RelocBatch-like.  Returns (cam_world [n,15], summaries, accepted, device_ms)."""
    o = default_options(**opt)
    n = batch.n_query
    ptr = np.ascontiguousarray(batch.match_ptr, dtype=np.int64)
    uvr = np.ascontiguousarray(batch.uv_ref, dtype=np.float32)
    uvc = np.ascontiguousarray(batch.uv_cur, dtype=np.float32)
    cref = np.ascontiguousarray(batch.cam_ref, dtype=np.float64)
    ccur = np.array(batch.cam_init, dtype=np.float64, order="C").copy()
    summ = (LmSummary * n)()
    acc = np.zeros(n, dtype=np.int32)
    ms = C.c_double()
    point_ptr = getattr(batch, "point_ptr", None)
    if point_ptr is not None:  # 2D-3D constraints per query (KRTOptimizer::Add2d3dConstraints), world points
        pptr = np.ascontiguousarray(point_ptr, dtype=np.int64)
        p2 = np.ascontiguousarray(batch.pts2d, dtype=np.float32)
        p3 = np.ascontiguousarray(batch.pts3d, dtype=np.float64)
        _check(lib().ptz_krt_solve_batch_2d3d(n, _p(ptr), _p(uvr), _p(uvc), _p(pptr), _p(p2), _p(p3), _p(cref), _p(ccur),
                                              batch.factor_type, C.c_double(max_reproj_error), C.byref(o), summ, _p(acc),
                                              C.byref(ms)),
               "ptz_krt_solve_batch_2d3d")
    else:
        _check(lib().ptz_krt_solve_batch(n, _p(ptr), _p(uvr), _p(uvc), _p(cref), _p(ccur), batch.factor_type,
                                         C.c_double(max_reproj_error), C.byref(o), summ, _p(acc), C.byref(ms)),
               "ptz_krt_solve_batch")
    return ccur, [s.as_dict() for s in summ], acc, ms.value


class KrtAttempt(C.Structure):
    _fields_ = [("table", C.c_void_p), ("entry", C.c_int32)]


class KrtTable:
    """ptz_krt_table: the matches of a rig's table entries, resident on a device (entry e owns matches
    [match_ptr[e], match_ptr[e+1]) of uv_ref / uv_cur)."""

    def __init__(self, match_ptr, uv_ref, uv_cur, device_id=0):
        ptr = np.ascontiguousarray(match_ptr, dtype=np.int64)
        uvr = np.ascontiguousarray(uv_ref, dtype=np.float32)
        uvc = np.ascontiguousarray(uv_cur, dtype=np.float32)
        self.n_entry = len(ptr) - 1
        self.handle = C.c_void_p()
        lib().ptz_krt_table_destroy.restype = None
        _check(lib().ptz_krt_table_create(self.n_entry, _p(ptr), _p(uvr), _p(uvc), int(device_id), C.byref(self.handle)), "ptz_krt_table_create")

    def close(self):
        if self.handle:
            lib().ptz_krt_table_destroy(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def krt_solve_attempts(attempts, cam_ref, cam_init, factor_type=0, max_reproj_error=100.0, **opt):
    """ptz_krt_solve_attempts: attempts = [(KrtTable, entry), ...].  Returns (cam_world [n,15], summaries, accepted, device_ms)."""
    o = default_options(**opt)
    n = len(attempts)
    att = (KrtAttempt * n)(*[KrtAttempt(t.handle.value, int(e)) for t, e in attempts])
    cref = np.ascontiguousarray(cam_ref, dtype=np.float64)
    ccur = np.array(cam_init, dtype=np.float64, order="C").copy()
    summ = (LmSummary * n)()
    acc = np.zeros(n, dtype=np.int32)
    ms = C.c_double()
    _check(lib().ptz_krt_solve_attempts(n, att, _p(cref), _p(ccur), int(factor_type), C.c_double(max_reproj_error), C.byref(o), summ, _p(acc),
                                        C.byref(ms)), "ptz_krt_solve_attempts")
    return ccur, [s.as_dict() for s in summ], acc, ms.value


def krt_solve_batch_device(n_query, d_match_ptr, d_uv_ref, d_uv_cur, d_cam_ref, d_cam_cur, d_summaries, d_accepted, factor_type=0,
                           max_reproj_error=100.0, d_point_ptr=None, d_pts2d=None, d_pts3d=None, stream=None, **opt):
    """ptz_krt_solve_batch_device: every d_* argument is a device buffer given as an object with .data_ptr() (a torch tensor)
    or as an integer address; d_summaries needs n_query * sizeof(LmSummary) bytes.  Enqueues on `stream` (integer hipStream_t
    handle, None = default stream) and returns without synchronising."""
    o = default_options(**opt)

    def ptr(x):
        if x is None:
            return None
        return C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))

    _check(lib().ptz_krt_solve_batch_device(int(n_query), ptr(d_match_ptr), ptr(d_uv_ref), ptr(d_uv_cur), ptr(d_point_ptr), ptr(d_pts2d),
                                            ptr(d_pts3d), ptr(d_cam_ref), ptr(d_cam_cur), int(factor_type), C.c_double(max_reproj_error),
                                            C.byref(o), ptr(d_summaries), ptr(d_accepted), C.c_void_p(stream) if stream else None),
           "ptz_krt_solve_batch_device")


def mfma_f64_peak(device_id=0):
    """Measured FP64 MFMA rate of the device in TFLOP/s (register-resident v_mfma_f64_16x16x4_f64 loop)."""
    t = C.c_double()
    _check(lib().ptz_mfma_f64_peak(int(device_id), C.byref(t)), "ptz_mfma_f64_peak")
    return t.value


def ba_solve_sharded(scenes, device_ids, **opt):
    """ptz_ba_solve_sharded: many scenes over several devices from this one process.  Returns (cams, rays, summaries)."""
    keep = []
    n = len(scenes)
    probs = (BaProblem * n)(*[_pack_problem(s, keep) for s in scenes])
    o = default_options(**opt)
    cam = np.ascontiguousarray(np.concatenate([s.cam_init for s in scenes]), dtype=np.float64)
    ray = np.ascontiguousarray(np.concatenate([s.ray_init for s in scenes]), dtype=np.float64)
    has_tlw = any(getattr(s, "tlw_init", None) is not None for s in scenes)
    tlw = np.ascontiguousarray(np.stack([getattr(s, "tlw_init", None) if getattr(s, "tlw_init", None) is not None else np.zeros(6)
                                         for s in scenes]), dtype=np.float64) if has_tlw else None
    dev = np.ascontiguousarray(device_ids, dtype=np.int32)
    summ = (LmSummary * n)()
    _check(lib().ptz_ba_solve_sharded(n, probs, _p(cam), _p(ray), _p(tlw) if tlw is not None else None, _p(dev), len(dev), C.byref(o), summ),
           "ptz_ba_solve_sharded")
    co = np.concatenate([[0], np.cumsum([s.n_cam for s in scenes])])
    ro = np.concatenate([[0], np.cumsum([s.n_ray for s in scenes])])
    cams = [cam[co[i]:co[i + 1]] for i in range(n)]
    rays = [ray[ro[i]:ro[i + 1]] for i in range(n)]
    return cams, rays, [s.as_dict() for s in summ]


def krt_solve_batch_sharded(batch, device_ids, max_reproj_error=100.0, **opt):
    """ptz_krt_solve_batch_sharded: the queries of `batch` over several devices from this one process."""
    o = default_options(**opt)
    n = batch.n_query
    ptr = np.ascontiguousarray(batch.match_ptr, dtype=np.int64)
    uvr = np.ascontiguousarray(batch.uv_ref, dtype=np.float32)
    uvc = np.ascontiguousarray(batch.uv_cur, dtype=np.float32)
    cref = np.ascontiguousarray(batch.cam_ref, dtype=np.float64)
    ccur = np.array(batch.cam_init, dtype=np.float64, order="C").copy()
    pp = p2 = p3 = None
    if getattr(batch, "point_ptr", None) is not None:
        pp = np.ascontiguousarray(batch.point_ptr, dtype=np.int64)
        p2 = np.ascontiguousarray(batch.pts2d, dtype=np.float32)
        p3 = np.ascontiguousarray(batch.pts3d, dtype=np.float64)
    dev = np.ascontiguousarray(device_ids, dtype=np.int32)
    summ = (LmSummary * n)()
    acc = np.zeros(n, dtype=np.int32)
    _check(lib().ptz_krt_solve_batch_sharded(n, _p(ptr), _p(uvr), _p(uvc), _p(pp) if pp is not None else None,
                                             _p(p2) if p2 is not None else None, _p(p3) if p3 is not None else None, _p(cref), _p(ccur),
                                             batch.factor_type, C.c_double(max_reproj_error), _p(dev), len(dev), C.byref(o), summ, _p(acc)),
           "ptz_krt_solve_batch_sharded")
    return ccur, [s.as_dict() for s in summ], acc


def hbm_bandwidth(device_id=0):
    """Measured HBM rates in GB/s: (streaming read, copy counted as read + write)."""
    r, c = C.c_double(), C.c_double()
    _check(lib().ptz_hbm_bandwidth(int(device_id), C.byref(r), C.byref(c)), "ptz_hbm_bandwidth")
    return r.value, c.value
