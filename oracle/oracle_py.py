"""ctypes binding of oracle/libptz_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package (ptz-calib_amd/) never does.  See oracle/ptz_oracle.h for the parity status.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2
JAC_NUMERIC, JAC_ANALYTIC = 0, 1


class LmOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("jacobian_mode", C.c_int32), ("num_threads", C.c_int32),
                ("reserved", C.c_int32), ("initial_trust_region_radius", C.c_double),
                ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
                ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double),
                ("max_lm_diagonal", C.c_double), ("function_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("max_num_consecutive_invalid_steps", C.c_int32), ("jacobi_scaling", C.c_int32)]


class LmSummary(C.Structure):
    _fields_ = [("termination_type", C.c_int32), ("num_iterations", C.c_int32), ("num_lm_steps", C.c_int32),
                ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
                ("num_residuals", C.c_int32), ("num_linear_solves", C.c_int32), ("num_jacobian_evals", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("final_radius", C.c_double),
                ("final_gradient_max_norm", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class LmTrace(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("count", C.c_int32), ("cost", C.POINTER(C.c_double)),
                ("cost_change", C.POINTER(C.c_double)), ("radius", C.POINTER(C.c_double)),
                ("rho", C.POINTER(C.c_double)), ("accepted", C.POINTER(C.c_int32))]


class BaProblem(C.Structure):
    _fields_ = [("n_cam", C.c_int32), ("n_ray", C.c_int32), ("n_obs", C.c_int64), ("obs_uv", C.c_void_p),
                ("obs_cam", C.c_void_p), ("obs_ray", C.c_void_p), ("ray_weight", C.c_void_p),
                ("n_obs3d", C.c_int32), ("obs3d_uv", C.c_void_p), ("obs3d_xyz", C.c_void_p),
                ("obs3d_cam", C.c_void_p), ("factor_type", C.c_int32), ("ic_of_cam", C.c_void_p)]


class KrtProblem(C.Structure):
    _fields_ = [("n_match", C.c_int32), ("uv_ref", C.c_void_p), ("uv_cur", C.c_void_p), ("cam_ref", C.c_void_p),
                ("factor_type", C.c_int32), ("n_pt", C.c_int32), ("pts2d", C.c_void_p), ("pts3d_local", C.c_void_p)]


def usable_cores() -> int:
    """CPU cores this process may actually use: min(affinity, cgroup quota).  The GPU box shows 256 logical
    CPUs but caps the container at a CFS quota; oversubscribed spinning OpenMP threads stall for minutes."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def build(force: bool = False) -> str:
    """Compile the oracle (and oracle/_ref when /root/reference is mounted).  Building the checker is
    not using it."""
    so = os.path.join(_HERE, "libptz_oracle.so")
    src = os.path.join(_HERE, "ptz_oracle.c")
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "std_sort_rank.cc")))
    if force or not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.check_call(["make", "-C", _HERE, "libptz_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/src/core/union_find.h"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libptz_oracle.so")
        if not os.path.exists(so):
            build()
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")  # before libgomp initialises
        os.environ.setdefault("GOMP_SPINCOUNT", "0")
        _LIB = C.CDLL(so)
        _LIB.orc_ba_cam_free_dim.restype = C.c_int32
        _LIB.orc_tracks_build.restype = C.c_int32
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_options(**kw) -> LmOptions:
    o = LmOptions()
    lib().orc_lm_options_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


@dataclass
class Trace:
    cost: np.ndarray
    cost_change: np.ndarray
    radius: np.ndarray
    rho: np.ndarray
    accepted: np.ndarray


def _mk_trace(cap):
    arrs = dict(cost=np.zeros(cap), cost_change=np.zeros(cap), radius=np.zeros(cap), rho=np.zeros(cap),
                accepted=np.zeros(cap, dtype=np.int32))
    t = LmTrace(cap, 0, arrs["cost"].ctypes.data_as(C.POINTER(C.c_double)),
                arrs["cost_change"].ctypes.data_as(C.POINTER(C.c_double)),
                arrs["radius"].ctypes.data_as(C.POINTER(C.c_double)),
                arrs["rho"].ctypes.data_as(C.POINTER(C.c_double)),
                arrs["accepted"].ctypes.data_as(C.POINTER(C.c_int32)))
    return t, arrs


def _ba_problem(obs_uv, obs_cam, obs_ray, ray_weight, n_cam, n_ray, factor_type, obs3d=None, ic_of_cam=None):
    keep = dict(obs_uv=np.ascontiguousarray(obs_uv, dtype=np.float32),
                obs_cam=np.ascontiguousarray(obs_cam, dtype=np.int32),
                obs_ray=np.ascontiguousarray(obs_ray, dtype=np.int32),
                ray_weight=np.ascontiguousarray(ray_weight, dtype=np.float64))
    p = BaProblem()
    p.n_cam, p.n_ray, p.n_obs = n_cam, n_ray, len(keep["obs_cam"])
    p.obs_uv, p.obs_cam, p.obs_ray, p.ray_weight = (_p(keep[k]) for k in ("obs_uv", "obs_cam", "obs_ray", "ray_weight"))
    p.factor_type = factor_type
    if obs3d is not None:
        keep["o3uv"] = np.ascontiguousarray(obs3d["uv"], dtype=np.float32)
        keep["o3xyz"] = np.ascontiguousarray(obs3d["xyz"], dtype=np.float64)
        keep["o3cam"] = np.ascontiguousarray(obs3d["cam"], dtype=np.int32)
        p.n_obs3d = len(keep["o3cam"])
        p.obs3d_uv, p.obs3d_xyz, p.obs3d_cam = _p(keep["o3uv"]), _p(keep["o3xyz"]), _p(keep["o3cam"])
    if ic_of_cam is not None:
        keep["ic"] = np.ascontiguousarray(ic_of_cam, dtype=np.int32)
        assert len(keep["ic"]) == n_cam
        p.ic_of_cam = _p(keep["ic"])
    return p, keep


def ba_solve(scene, cam0=None, ray0=None, tlw0=None, obs3d=None, trace=False, disp=None, **opt):
    """Run the oracle's PTZ-IBA LM.  Returns (cam, ray, tlw, summary dict, Trace|None).
    disp: optional float64[3] array, the displacement block of PTZRayDistDisp -- initial value in, refined value out
    (updated in place); None starts it at zero as the reference does and drops the result."""
    p, keep = _ba_problem(scene.obs_uv, scene.obs_cam, scene.obs_ray, scene.ray_weight, scene.n_cam, scene.n_ray,
                          scene.factor_type, obs3d, getattr(scene, "ic_of_cam", None))
    cam = np.array(scene.cam_init if cam0 is None else cam0, dtype=np.float64, order="C").copy()
    ray = np.array(scene.ray_init if ray0 is None else ray0, dtype=np.float64, order="C").copy()
    tlw = np.zeros(6) if tlw0 is None else np.array(tlw0, dtype=np.float64).copy()
    o = default_options(**opt)
    o.num_threads = max(1, min(o.num_threads, usable_cores()))
    s = LmSummary()
    t, arrs = _mk_trace(o.max_num_iterations + 2) if trace else (None, None)
    if disp is not None:
        assert disp.dtype == np.float64 and disp.shape == (3,) and disp.flags.c_contiguous
    rc = lib().orc_ba_solve_disp(C.byref(p), _p(cam), _p(ray), _p(tlw), _p(disp) if disp is not None else None, C.byref(o),
                                 C.byref(s), C.byref(t) if t is not None else None)
    if rc != 0:
        raise RuntimeError("orc_ba_solve: invalid problem")
    tr = None
    if t is not None:
        n = t.count
        tr = Trace(**{k: v[:n].copy() for k, v in arrs.items()})
    return cam, ray, tlw, s.as_dict(), tr


def ba_linearize(scene, cam, ray, tlw=None, jacobian_mode=JAC_ANALYTIC, obs3d=None, disp=None):
    p, keep = _ba_problem(scene.obs_uv, scene.obs_cam, scene.obs_ray, scene.ray_weight, scene.n_cam, scene.n_ray,
                          scene.factor_type, obs3d)
    ncf = lib().orc_ba_cam_free_dim(scene.factor_type)
    cam = np.ascontiguousarray(cam, dtype=np.float64)
    ray = np.ascontiguousarray(ray, dtype=np.float64)
    tlw = np.zeros(6) if tlw is None else np.ascontiguousarray(tlw, dtype=np.float64)
    cost = C.c_double()
    g_c = np.zeros((scene.n_cam, ncf))
    U = np.zeros((scene.n_cam, ncf, ncf))
    g_r = np.zeros((scene.n_ray, 3))
    V = np.zeros((scene.n_ray, 3, 3))
    W = np.zeros((p.n_obs, ncf, 3))
    dd = None if disp is None else np.ascontiguousarray(disp, dtype=np.float64)
    rc = lib().orc_ba_linearize_disp(C.byref(p), _p(cam), _p(ray), _p(tlw), _p(dd) if dd is not None else None, jacobian_mode,
                                     C.byref(cost), _p(g_c), _p(U), _p(g_r), _p(V), _p(W))
    if rc != 0:
        raise RuntimeError("orc_ba_linearize: invalid problem")
    return dict(cost=cost.value, g_c=g_c, U=U, g_r=g_r, V=V, W=W, ncf=ncf)


def ba_residuals(scene, cam, ray, tlw=None, obs3d=None, disp=None):
    p, keep = _ba_problem(scene.obs_uv, scene.obs_cam, scene.obs_ray, scene.ray_weight, scene.n_cam, scene.n_ray,
                          scene.factor_type, obs3d)
    cam = np.ascontiguousarray(cam, dtype=np.float64)
    ray = np.ascontiguousarray(ray, dtype=np.float64)
    tlw = np.zeros(6) if tlw is None else np.ascontiguousarray(tlw, dtype=np.float64)
    res = np.zeros((p.n_obs + p.n_obs3d, 2))
    dd = None if disp is None else np.ascontiguousarray(disp, dtype=np.float64)
    lib().orc_ba_residuals_disp(C.byref(p), _p(cam), _p(ray), _p(tlw), _p(dd) if dd is not None else None, _p(res))
    return res


def pix2ray(scene, cam):
    p, keep = _ba_problem(scene.obs_uv, scene.obs_cam, scene.obs_ray, scene.ray_weight, scene.n_cam, scene.n_ray,
                          scene.factor_type)
    cam = np.ascontiguousarray(cam, dtype=np.float64)
    ray = np.zeros((scene.n_ray, 3))
    lib().orc_pix2ray(C.byref(p), _p(cam), _p(ray))
    return ray


def krt_world_to_local(cam_ref, cam_cur):
    out = np.zeros(15)
    lib().orc_krt_world_to_local(_p(np.ascontiguousarray(cam_ref, dtype=np.float64)),
                                 _p(np.ascontiguousarray(cam_cur, dtype=np.float64)), _p(out))
    return out


def krt_local_to_world(cam_ref, cam_loc, factor_type):
    out = np.zeros(15)
    lib().orc_krt_local_to_world(_p(np.ascontiguousarray(cam_ref, dtype=np.float64)),
                                 _p(np.ascontiguousarray(cam_loc, dtype=np.float64)), factor_type, _p(out))
    return out


def krt_point_to_local(cam_ref, pts3d_world):
    """R_local_world X_w + t_local_world (krt_optimizer.cc:357-362)."""
    cam_ref = np.ascontiguousarray(cam_ref, dtype=np.float64)
    X = np.ascontiguousarray(pts3d_world, dtype=np.float64).reshape(-1, 3)
    out = np.zeros_like(X)
    for i in range(len(X)):
        lib().orc_krt_point_to_local(_p(cam_ref), _p(X[i]), _p(out[i]))
    return out


def res_2d3d_krt(cam_local, fxfy, pt2d, pt3d_local):
    cam = np.ascontiguousarray(cam_local, dtype=np.float64)
    a = np.ascontiguousarray(pt2d, dtype=np.float32)
    X = np.ascontiguousarray(pt3d_local, dtype=np.float64)
    res = np.zeros(2)
    lib().orc_res_2d3d_krt(_p(cam), C.c_int32(int(fxfy)), _p(a), _p(X), _p(res))
    return res


def krt_solve(uv_ref, uv_cur, cam_ref, cam_cur_local, factor_type=0, trace=False, pts2d=None, pts3d_local=None, **opt):
    """Single-view LM in the reference camera's local frame.  Returns (cam_local, summary, Trace|None).
    pts2d / pts3d_local: optional 2D-3D constraints (Add2d3dConstraints), points already in the local frame."""
    uv_ref = np.ascontiguousarray(uv_ref, dtype=np.float32)
    uv_cur = np.ascontiguousarray(uv_cur, dtype=np.float32)
    cam_ref = np.ascontiguousarray(cam_ref, dtype=np.float64)
    cam = np.array(cam_cur_local, dtype=np.float64).copy()
    if pts2d is not None and len(pts2d):
        pts2d = np.ascontiguousarray(pts2d, dtype=np.float32)
        pts3d_local = np.ascontiguousarray(pts3d_local, dtype=np.float64)
        p = KrtProblem(len(uv_ref), _p(uv_ref), _p(uv_cur), _p(cam_ref), factor_type, len(pts2d), _p(pts2d), _p(pts3d_local))
    else:
        p = KrtProblem(len(uv_ref), _p(uv_ref), _p(uv_cur), _p(cam_ref), factor_type, 0, None, None)
    o = default_options(**opt)
    s = LmSummary()
    t, arrs = _mk_trace(o.max_num_iterations + 2) if trace else (None, None)
    lib().orc_krt_solve(C.byref(p), _p(cam), C.byref(o), C.byref(s), C.byref(t) if t is not None else None)
    tr = None
    if t is not None:
        tr = Trace(**{k: v[: t.count].copy() for k, v in arrs.items()})
    return cam, s.as_dict(), tr


def krt_solve_batch(rb, n_query=None, max_reproj_error=100.0, num_threads=1, **opt):
    """run_ptz_reloc.cc:68-118 over the first n_query queries of a packed RelocBatch (synth.make_reloc_queries): returns
    (cam_world [n, 15], list of summaries, accepted [n]).  num_threads deals whole queries to threads."""
    n = rb.n_query if n_query is None else min(int(n_query), rb.n_query)
    ptr = np.ascontiguousarray(rb.match_ptr[: n + 1], dtype=np.int64)
    uv_ref = np.ascontiguousarray(rb.uv_ref, dtype=np.float32)
    uv_cur = np.ascontiguousarray(rb.uv_cur, dtype=np.float32)
    cam_ref = np.ascontiguousarray(rb.cam_ref[:n], dtype=np.float64)
    cam = np.array(rb.cam_init[:n], dtype=np.float64).copy()
    o = default_options(**opt)
    summ = (LmSummary * n)()
    acc = np.zeros(n, dtype=np.int32)
    lib().orc_krt_solve_batch(C.c_int32(n), _p(ptr), _p(uv_ref), _p(uv_cur), _p(cam_ref), _p(cam), C.c_int32(int(rb.factor_type)),
                              C.c_double(max_reproj_error), C.byref(o), summ, _p(acc), C.c_int32(int(num_threads)))
    return cam, [s.as_dict() for s in summ], acc


def krt_check(summary: dict, cam_local, max_reproj_error: float) -> bool:
    s = LmSummary()
    for k, v in summary.items():
        setattr(s, k, v)
    return bool(lib().orc_krt_check(C.byref(s), _p(np.ascontiguousarray(cam_local, dtype=np.float64)),
                                    C.c_double(max_reproj_error)))


def rodrigues(rvec):
    R = np.zeros(9)
    lib().orc_rodrigues(_p(np.ascontiguousarray(rvec, dtype=np.float64)), _p(R))
    return R.reshape(3, 3)


def rodrigues_jac(rvec):
    R = np.zeros(9)
    dR = np.zeros(27)
    lib().orc_rodrigues_jac(_p(np.ascontiguousarray(rvec, dtype=np.float64)), _p(R), _p(dR))
    return R.reshape(3, 3), dR.reshape(3, 3, 3)


def rodrigues_inv(R):
    r = np.zeros(3)
    lib().orc_rodrigues_inv(_p(np.ascontiguousarray(R, dtype=np.float64).reshape(9)), _p(r))
    return r


def tracks_build(pairs, min_track_length=4):
    """pairs: list of (src, dst, [(queryIdx, trainIdx), ...]).  Returns {track_id: {image: feature}}."""
    n = len(pairs)
    src = np.array([p[0] for p in pairs], dtype=np.int64)
    dst = np.array([p[1] for p in pairs], dtype=np.int64)
    ptr = np.zeros(n + 1, dtype=np.int64)
    for i, p in enumerate(pairs):
        ptr[i + 1] = ptr[i] + len(p[2])
    q = np.array([m[0] for p in pairs for m in p[2]], dtype=np.int32)
    t = np.array([m[1] for p in pairs for m in p[2]], dtype=np.int32)
    tid = C.POINTER(C.c_int32)()
    tptr = C.POINTER(C.c_int64)()
    eimg = C.POINTER(C.c_int32)()
    efeat = C.POINTER(C.c_int32)()
    nt = lib().orc_tracks_build(n, _p(src), _p(dst), _p(ptr), _p(q), _p(t), min_track_length, C.byref(tid),
                                C.byref(tptr), C.byref(eimg), C.byref(efeat))
    out = {}
    for k in range(nt):
        out[int(tid[k])] = {int(eimg[e]): int(efeat[e]) for e in range(tptr[k], tptr[k + 1])}
    for ptr_ in (tid, tptr, eimg, efeat):
        lib().orc_free(ptr_)
    return out


def rank_by_score(score):
    """Image ids with a positive score, best first, ordered by the toolchain's std::sort (oracle/std_sort_rank.cc)."""
    score = np.ascontiguousarray(score, dtype=np.float32)
    out = np.zeros(len(score), dtype=np.int64)
    m = lib().orc_rank_by_score(len(score), _p(score), _p(out))
    return [int(x) for x in out[:m]]


def ref_tracks_available() -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", "ref_tracks"))


def ref_tracks_build(pairs, min_track_length=4):
    """Same, through oracle/_ref/ref_tracks (the reference's own union_find.h / flat_pair_map.h)."""
    lines = [str(len(pairs))]
    for s_, d_, ms in pairs:
        lines.append(f"{s_} {d_} {len(ms)}")
        lines.extend(f"{a} {b}" for a, b in ms)
    lines.append(str(min_track_length))
    out = subprocess.run([os.path.join(_HERE, "_ref", "ref_tracks")], input="\n".join(lines) + "\n",
                         capture_output=True, text=True, check=True).stdout.split()
    it = iter(out)
    nt = int(next(it))
    res = {}
    for _ in range(nt):
        tid, ne = int(next(it)), int(next(it))
        res[tid] = {}
        for _ in range(ne):
            img, feat = int(next(it)), int(next(it))
            res[tid][img] = feat
    return res
