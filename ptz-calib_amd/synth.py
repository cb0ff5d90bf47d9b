"""Deterministic synthetic PTZ rigs and relocalization queries.

The reference's datasets (README.md:25) are not available, so every BASELINE config is synthetic.
Generation is bit-reproducible across machines: SplitMix64 counters + Box-Muller, no numpy RNG.

Scene s uses seed 0x50545A00 + s.  Output is the packed form the C-ABI takes (include/ptz_calib_amd.h):
observations sorted (track asc, image asc) as PTZRayOptimizer::AddConstraints2d2d iterates them
(ptzray_optimizer.cc:801-850), float32 pixels (data_io.cc:40), camera 15-vectors (types.cc:32-73).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

SEED_BASE = 0x50545A00
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


_SYNTH_LIB = False


def _synth_lib(name: str = "ptz_synth_project"):
    """A function of libptzsynth.so (host/synth_kernels.c, built by __graft_entry__.build() / host/Makefile), or None: numpy does the
    same arithmetic (PTZ_SYNTH_NUMPY=1 forces that; tests hold both paths to the same scene hashes)."""
    global _SYNTH_LIB
    if _SYNTH_LIB is False:
        import ctypes
        import os
        _SYNTH_LIB = None
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libptzsynth.so")
        if os.environ.get("PTZ_SYNTH_NUMPY") != "1" and os.path.exists(path):
            try:
                dll = ctypes.CDLL(path)
                fn = dll.ptz_synth_project
                fn.restype = None
                fn.argtypes = [ctypes.c_int64] + [ctypes.c_void_p] * 6 + [ctypes.c_double] * 4 + [ctypes.c_void_p] * 3
                fc = dll.ptz_synth_candidates
                fc.restype = None
                fc.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 6
                _SYNTH_LIB = {"ptz_synth_project": fn, "ptz_synth_candidates": fc}
            except (OSError, AttributeError):
                _SYNTH_LIB = None
    return _SYNTH_LIB.get(name) if _SYNTH_LIB else None


def usable_cores() -> int:
    """CPU cores this process may actually use: min(affinity, cgroup CPU quota).  A GPU box shows hundreds of logical CPUs and caps
    the container at a CFS quota; pools sized by os.cpu_count() oversubscribe it several times over (eight ranks x 16 workers)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


class SplitMix64:
    """Counter-based SplitMix64 stream; draw(n) returns the next n outputs."""

    def __init__(self, seed: int):
        self.seed = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        self.count = 0

    def u64(self, n: int) -> np.ndarray:
        idx = np.arange(self.count + 1, self.count + n + 1, dtype=np.uint64)
        self.count += n
        with np.errstate(over="ignore"):
            z = self.seed + idx * _GOLDEN
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            z = z ^ (z >> np.uint64(31))
        return z

    def uniform(self, n: int, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
        u = (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return lo + (hi - lo) * u

    def normal(self, n: int, sigma: float = 1.0) -> np.ndarray:
        m = (n + 1) // 2
        u1 = 1.0 - self.uniform(m)  # (0, 1]
        u2 = self.uniform(m)
        r = np.sqrt(-2.0 * np.log(u1))
        z = np.concatenate([r * np.cos(2.0 * math.pi * u2), r * np.sin(2.0 * math.pi * u2)])[:n]
        return sigma * z

    def permutation_keys(self, n: int) -> np.ndarray:
        return self.u64(n)

    def u64_at(self, offsets: np.ndarray) -> np.ndarray:
        """The outputs u64() would return at positions count + 1 + offsets of the stream, without moving it."""
        idx = np.uint64(self.count + 1) + np.asarray(offsets).astype(np.uint64)
        with np.errstate(over="ignore"):
            z = self.seed + idx * _GOLDEN
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            z = z ^ (z >> np.uint64(31))
        return z

    def skip(self, n: int) -> None:
        self.count += n


def rodrigues(rvec: np.ndarray) -> np.ndarray:
    """Vector -> rotation matrix (same formula as cv::Rodrigues)."""
    rvec = np.asarray(rvec, dtype=np.float64)
    theta = float(np.linalg.norm(rvec))
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    c, s = math.cos(theta), math.sin(theta)
    r = rvec / theta
    rx = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    return c * np.eye(3) + (1 - c) * np.outer(r, r) + s * rx


def rodrigues_inv(R: np.ndarray) -> np.ndarray:
    """Rotation matrix -> vector (angle in [0, pi])."""
    R = np.asarray(R, dtype=np.float64)
    c = max(-1.0, min(1.0, (np.trace(R) - 1.0) * 0.5))
    theta = math.acos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = 0.5 * np.linalg.norm(v)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = np.sqrt(np.maximum((np.diag(R) + 1) * 0.5, 0.0))
        t[1] *= -1.0 if R[0, 1] < 0 else 1.0
        t[2] *= -1.0 if R[0, 2] < 0 else 1.0
        if abs(t[0]) < abs(t[1]) and abs(t[0]) < abs(t[2]) and (R[1, 2] > 0) != (t[1] * t[2] > 0):
            t[2] = -t[2]
        return t * (theta / np.linalg.norm(t))
    return v * (theta / (2.0 * s))


def _rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _azimuth_window(width: int, f_min: float, distorted: bool, tilt_max: float) -> float:
    """Half-width (radians) of the azimuth window outside which no ray of the +-35 degree elevation band can project into a
    frame: with d = az + pan, the camera-frame point is (cos el sin d, .., ..) before the tilt (which leaves x alone) and
    |z| <= 1, so u in [8, width - 8] needs cos el |sin d| <= x_lim = (width / 2 - 8) / (f_min rad_min), rad_min = 1 - 0.05 * 1.5
    with distortion (r2 < 1.5 is part of the visibility test).  Behind the camera (|d| > 90 degrees) z is negative as long as
    the tilt stays small; if any of this does not hold the window is the full circle (the dense evaluation)."""
    x_lim = (0.5 * width - 8.0) / (f_min * (0.925 if distorted else 1.0))
    s_lim = x_lim / math.cos(math.radians(35.0))
    if s_lim >= 0.94 or tilt_max > math.radians(12.0):
        return math.pi
    return math.asin(s_lim) + math.radians(0.5)


def _candidate_pairs(az: np.ndarray, pan: np.ndarray, window: float):
    """(view, ray) index pairs with |az + pan| (mod 2 pi) <= window, view-major, rays ascending inside a view."""
    n_ray = len(az)
    if window >= math.pi:
        return np.repeat(np.arange(len(pan)), n_ray), np.tile(np.arange(n_ray), len(pan))
    two_pi = 2.0 * math.pi
    a = np.mod(az, two_pi)
    idx = np.argsort(a, kind="stable")
    a_sorted = a[idx]
    cand = _synth_lib("ptz_synth_candidates")
    if cand is not None:
        # the three slices of every view by vectorised searchsorted, their union in ascending ray order by a bitmap in C
        # (host/synth_kernels.c) instead of a Python loop over the views with a sort each: the same pairs in the same order
        c = np.array([(-pn) % two_pi for pn in pan], dtype=np.float64)
        lo, hi = c - window, c + window
        n_v = len(pan)
        L = np.zeros((n_v, 3), dtype=np.int64); H = np.zeros((n_v, 3), dtype=np.int64)
        wl, wh = lo < 0.0, hi > two_pi
        L[wl, 0] = np.searchsorted(a_sorted, lo[wl] + two_pi, "left"); H[wl, 0] = n_ray
        H[wh, 1] = np.searchsorted(a_sorted, hi[wh] - two_pi, "right")
        lo_c, hi_c = np.where(wl, 0.0, lo), np.where(wh, two_pi, hi)
        L[:, 2] = np.searchsorted(a_sorted, lo_c, "left"); H[:, 2] = np.searchsorted(a_sorted, hi_c, "right")
        H = np.maximum(H, L)
        counts = (H - L).sum(axis=1)
        offs = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
        total = int(counts.sum())
        vi = np.empty(total, dtype=np.int64); pi_ = np.empty(total, dtype=np.int64)
        bitmap = np.zeros((n_ray + 63) // 64, dtype=np.uint64)
        idx64 = np.ascontiguousarray(idx, dtype=np.int64); L = np.ascontiguousarray(L); H = np.ascontiguousarray(H)
        cand(n_ray, idx64.ctypes.data, n_v, L.ctypes.data, H.ctypes.data, offs.ctypes.data, bitmap.ctypes.data, vi.ctypes.data, pi_.ctypes.data)
        return vi, pi_
    vi, pi_ = [], []
    for i, pn in enumerate(pan):
        c = (-pn) % two_pi  # the azimuth this view looks at
        lo, hi = c - window, c + window
        parts = []
        if lo < 0.0:
            parts.append(idx[np.searchsorted(a_sorted, lo + two_pi, "left"):])
            lo = 0.0
        if hi > two_pi:
            parts.append(idx[: np.searchsorted(a_sorted, hi - two_pi, "right")])
            hi = two_pi
        parts.append(idx[np.searchsorted(a_sorted, lo, "left"): np.searchsorted(a_sorted, hi, "right")])
        p = np.sort(np.concatenate(parts))
        vi.append(np.full(len(p), i, dtype=np.int64))
        pi_.append(p)
    return np.concatenate(vi), np.concatenate(pi_)


@dataclass
class Scene:
    """One packed PTZ-IBA problem + ground truth."""

    n_cam: int
    n_ray: int
    width: int
    height: int
    factor_type: int
    obs_uv: np.ndarray       # float32 [n_obs, 2]
    obs_cam: np.ndarray      # int32 [n_obs]
    obs_ray: np.ndarray      # int32 [n_obs] non-decreasing
    ray_weight: np.ndarray   # float64 [n_ray]  (= track length)
    cam_gt: np.ndarray       # float64 [n_cam, 15]
    cam_init: np.ndarray     # float64 [n_cam, 15]
    ray_gt: np.ndarray       # float64 [n_ray, 3]
    ray_init: np.ndarray     # float64 [n_ray, 3] (Pix2Ray of cam_init)
    seed: int = 0
    meta: dict = field(default_factory=dict)
    obs3d: dict | None = None        # 2D-3D annotations: {uv float32 [m,2], xyz float64 [m,3], cam int32 [m]}
    tlw_gt: np.ndarray | None = None   # T_l_w as [rvec, t] (ptzray_optimizer.cc:507-513)
    tlw_init: np.ndarray | None = None
    ic_of_cam: np.ndarray | None = None  # int32 [n_cam] intrinsics-block id per camera (SetSharedIntrinsics); None = own block

    @property
    def n_obs(self) -> int:
        return int(self.obs_cam.shape[0])


def pix2ray(obs_uv, obs_cam, obs_ray, n_ray, cam):
    """Pix2Ray, ptzray_optimizer.cc:768-797 (vectorised; same formula as oracle orc_pix2ray)."""
    R = np.stack([rodrigues(c[4:7]) for c in cam])  # [n_cam,3,3]
    q = np.stack([
        (obs_uv[:, 0].astype(np.float64) - cam[obs_cam, 2]) / cam[obs_cam, 0],
        (obs_uv[:, 1].astype(np.float64) - cam[obs_cam, 3]) / cam[obs_cam, 1],
        np.ones(len(obs_cam)),
    ], axis=1)
    Rinv = np.linalg.inv(R)[obs_cam]
    t = np.einsum("nij,nj->ni", Rinv, q)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    acc = np.zeros((n_ray, 3))
    np.add.at(acc, obs_ray, t)
    cnt = np.bincount(obs_ray, minlength=n_ray).astype(np.float64)
    acc /= cnt[:, None]
    acc /= np.linalg.norm(acc, axis=1, keepdims=True)
    return acc


def make_scene(scene_id: int = 0, n_views: int = 200, obs_per_view: int = 500, factor_type: int = 0,
               width: int = 1920, height: int = 1080, pan_range_deg: float | None = None,
               noise_px: float = 0.5, init_rot_sigma_deg: float = 0.5, init_focal: float | None = None,
               min_track_len: int = 4, n_intrinsics_groups: int | None = None) -> Scene:
    """Synthetic PTZ rig.  C1: n_views=20, obs_per_view=100.  C2: 200 x 500.

    pan_range_deg=None picks min(360, 6 * n_views): a full 360-degree ring needs ~1.8-degree pan spacing for
    tracks of >= 4 views (Filter(4), ptzray_optimizer.cc:541) to exist, so small rigs sweep a sector instead."""
    if pan_range_deg is None:
        pan_range_deg = min(360.0, 6.0 * n_views)
    seed = SEED_BASE + scene_id
    rng = SplitMix64(seed)
    cx, cy = 0.5 * width, 0.5 * height
    N = n_views
    # --- ground-truth cameras
    pan = np.deg2rad(pan_range_deg * np.arange(N) / N + rng.uniform(N, -0.2, 0.2))
    if pan_range_deg < 360.0:
        pan -= np.deg2rad(pan_range_deg) * 0.5
    tilt = np.deg2rad(np.array([-10.0, 0.0, 10.0])[np.arange(N) % 3] + rng.uniform(N, -0.2, 0.2))
    focal = rng.uniform(N, 1800.0, 3200.0) * (width / 1920.0)
    k1 = rng.uniform(N, -0.05, 0.05) if factor_type != 0 else np.zeros(N)
    ic_of_cam = None
    if n_intrinsics_groups is not None:
        # fixed-zoom rigs: views i with the same (i mod G) share one intrinsics block (SetSharedIntrinsics)
        ic_of_cam = (np.arange(N) % n_intrinsics_groups).astype(np.int32)
        focal = focal[:n_intrinsics_groups][ic_of_cam]
        k1 = k1[:n_intrinsics_groups][ic_of_cam]
    Rgt = np.stack([_rot_x(tilt[i]) @ _rot_y(pan[i]) for i in range(N)])
    cam_gt = np.zeros((N, 15))
    cam_gt[:, 0] = focal
    cam_gt[:, 1] = focal
    cam_gt[:, 2] = cx
    cam_gt[:, 3] = cy
    cam_gt[:, 10] = k1
    for i in range(N):
        cam_gt[i, 4:7] = rodrigues_inv(Rgt[i])

    # --- candidate world rays, uniform in the band |elevation| <= 35 deg.  The ray count P is sized
    #     (deterministic fixed point) so that ~2.2x the target observations are visible: keeping ~45 %
    #     of them per view then gives tracks of mean length ~5 after Filter(4).
    # Only (view, ray) pairs that CAN be visible are ever evaluated: a ray at azimuth az is seen by a view at pan angle pan
    # only if |az + pan| (mod 360) is below a bound that follows from the frame width, the shortest focal length and the
    # elevation band (derivation at _azimuth_window).  The arithmetic per pair is the dense generator's, element for element
    # (round 2 evaluated all N x P pairs, 2.4 s per C2 scene; tests/golden/scene_hashes.json holds this one to its bits).
    target_total = N * obs_per_view
    half = math.pi if pan_range_deg >= 360.0 else math.radians(0.5 * pan_range_deg + 30.0)
    window = _azimuth_window(width, float(focal.min()), factor_type != 0, float(np.abs(tilt).max()))
    P = int(round(target_total / 3.2))
    for _attempt in range(4):
        az = rng.uniform(P, -half, half)
        sin_el = rng.uniform(P, -math.sin(math.radians(35.0)), math.sin(math.radians(35.0)))
        cos_el = np.sqrt(1.0 - sin_el ** 2)
        X = np.stack([cos_el * np.sin(az), sin_el, cos_el * np.cos(az)], axis=1)  # unit, z forward at pan 0
        vi, pi_ = _candidate_pairs(az, pan, window)  # view-major, rays ascending inside a view
        # exact projections and visibility (>= 8 px inside the frame)
        proj = _synth_lib()
        if proj is not None:
            # the same arithmetic in C (host/synth_kernels.c, -ffp-contract=off: every product and sum rounded where numpy rounds it)
            vi = np.ascontiguousarray(vi, dtype=np.int64); pi_ = np.ascontiguousarray(pi_, dtype=np.int64)
            u = np.empty(len(vi)); v = np.empty(len(vi)); vis8 = np.empty(len(vi), dtype=np.uint8)
            Rflat = np.ascontiguousarray(Rgt.reshape(N, 9)); Xc = np.ascontiguousarray(X)
            fc = np.ascontiguousarray(focal, dtype=np.float64); kc = np.ascontiguousarray(k1, dtype=np.float64)
            proj(len(vi), vi.ctypes.data, pi_.ctypes.data, Rflat.ctypes.data, Xc.ctypes.data, fc.ctypes.data, kc.ctypes.data,
                 float(cx), float(cy), float(width), float(height), u.ctypes.data, v.ctypes.data, vis8.ctypes.data)
            vis = vis8.view(np.bool_)
            ratio = 2.2 * target_total / max(int(vis.sum()), 1)
            if 0.92 < ratio < 1.08 or _attempt == 3:
                break
            P = max(16, int(round(P * ratio)))
            continue
        # (the pairs are view-major: a camera's entries are np.repeat of its value over its run of pairs -- the gathers'
        #  values, at a third of their cost; the rays' coordinates are taken from contiguous columns)
        runs = np.bincount(vi, minlength=N)
        per_view = lambda a: np.repeat(a, runs)  # noqa: E731
        X0, X1, X2 = np.take(np.ascontiguousarray(X[:, 0]), pi_), np.take(np.ascontiguousarray(X[:, 1]), pi_), np.take(np.ascontiguousarray(X[:, 2]), pi_)
        Pc = [per_view(Rgt[:, i, 0]) * X0 + per_view(Rgt[:, i, 1]) * X1 + per_view(Rgt[:, i, 2]) * X2 for i in range(3)]
        z = Pc[2]
        with np.errstate(divide="ignore", invalid="ignore"):
            x = Pc[0] / z
            y = Pc[1] / z
            r2 = x * x + y * y
            rad = 1.0 + per_view(k1) * r2
            fv = per_view(focal)
            u = fv * x * rad + cx
            v = fv * y * rad + cy
        vis = (z > 0.1) & (u >= 8) & (u <= width - 8) & (v >= 8) & (v <= height - 8) & (r2 < 1.5)
        ratio = 2.2 * target_total / max(int(vis.sum()), 1)
        if 0.92 < ratio < 1.08 or _attempt == 3:
            break
        P = max(16, int(round(P * ratio)))

    # --- per view keep a seeded random subset; tune the per-view quota so that, after dropping
    #     tracks shorter than min_track_len, the mean is ~obs_per_view (deterministic fixed point)
    # (the dense generator drew one key per (view, ray) cell, cell (i, p) the (i P + p + 1)-th of the stream: the visible
    #  cells' keys are evaluated at those positions and the stream moves on by N P)
    vi, pi_, u, v = vi[vis], pi_[vis], u[vis], v[vis]
    keys = rng.u64_at(vi.astype(np.int64) * P + pi_) >> np.uint64(1)
    rng.skip(N * P)
    order = np.lexsort((pi_, keys, vi))  # per view: ascending key, ties by ray number (as a stable argsort of the dense row)
    vi, pi_, u, v = vi[order], pi_[order], u[order], v[order]
    nvis = np.bincount(vi, minlength=N)
    first = np.concatenate([[0], np.cumsum(nvis)[:-1]])
    rank = np.arange(len(vi)) - first[vi]  # position of the pair in its view's random order
    quota = float(obs_per_view) * 1.15
    sel = None
    for _ in range(6):
        q = np.minimum(nvis, int(round(quota)))
        sel = rank < q[vi]
        tl = np.bincount(pi_[sel], minlength=P)
        sel &= tl[pi_] >= min_track_len
        mean_obs = sel.sum() / N
        if abs(mean_obs - obs_per_view) < 0.01 * obs_per_view:
            break
        quota *= obs_per_view / max(mean_obs, 1.0)
    vi, pi_, u, v = vi[sel], pi_[sel], u[sel], v[sel]
    ray_ids = np.unique(pi_)
    n_ray = len(ray_ids)
    # observation list sorted (ray asc, cam asc)
    o = np.lexsort((vi, pi_))
    cc, rr = vi[o], np.searchsorted(ray_ids, pi_[o])
    n_obs = len(rr)
    uu = u[o] + rng.normal(n_obs, noise_px)
    vv = v[o] + rng.normal(n_obs, noise_px)
    obs_uv = np.stack([uu, vv], axis=1).astype(np.float32)
    obs_cam = cc.astype(np.int32)
    obs_ray = rr.astype(np.int32)
    ray_weight = np.bincount(obs_ray, minlength=n_ray).astype(np.float64)
    ray_gt = X[ray_ids]

    # --- initial guess as the reference would hand it to the global BA
    f0 = init_focal if init_focal is not None else 1.2 * max(width, height)  # ptz_incremental_optimizer.cc:324
    cam_init = cam_gt.copy()
    cam_init[:, 0] = f0
    cam_init[:, 1] = f0
    cam_init[:, 10] = 0.0
    sig = math.radians(init_rot_sigma_deg)
    pert = rng.normal(3 * N, sig).reshape(N, 3)
    for i in range(N):
        cam_init[i, 4:7] = rodrigues_inv(rodrigues(pert[i]) @ Rgt[i])
    ray_init = pix2ray(obs_uv, obs_cam, obs_ray, n_ray, cam_init)
    return Scene(n_cam=N, n_ray=n_ray, width=width, height=height, factor_type=factor_type, obs_uv=obs_uv,
                 obs_cam=obs_cam, obs_ray=obs_ray, ray_weight=ray_weight, cam_gt=cam_gt, cam_init=cam_init,
                 ray_gt=ray_gt, ray_init=ray_init, seed=seed, ic_of_cam=ic_of_cam,
                 meta={"obs_per_view_mean": n_obs / N, "track_len_mean": n_obs / max(n_ray, 1)})


@dataclass
class RelocBatch:
    """Batched single-view relocalization queries (run_ptz_reloc.cc:68-118) in packed form."""

    n_query: int
    match_ptr: np.ndarray   # int64 [n_query + 1]
    uv_ref: np.ndarray      # float32 [n_match_total, 2]
    uv_cur: np.ndarray      # float32 [n_match_total, 2]
    cam_ref: np.ndarray     # float64 [n_query, 15]  reference camera of each query (world frame)
    cam_init: np.ndarray    # float64 [n_query, 15]  initial current camera, world frame (run_ptz_reloc.cc:96-104)
    cam_gt: np.ndarray      # float64 [n_query, 15]
    factor_type: int = 0
    # optional 2D-3D constraints per query (KRTOptimizer::Add2d3dConstraints, krt_optimizer.cc:350-383)
    point_ptr: np.ndarray | None = None   # int64 [n_query + 1]
    pts2d: np.ndarray | None = None       # float32 [n_pt_total, 2]
    pts3d: np.ndarray | None = None       # float64 [n_pt_total, 3]  world points


def make_reloc_batch(n_query: int, n_match: int = 128, seed_id: int = 0, factor_type: int = 0,
                     rig: np.ndarray | None = None, width: int = 1920, height: int = 1080,
                     noise_px: float = 0.5) -> RelocBatch:
    """C5: queries with pan/tilt over the covered band, f ~ U(1500, 4000), n_match matches against the
    nearest reference view, 0.5 px noise, init = (f_ref, R_ref)."""
    rng = SplitMix64(SEED_BASE + 0x100000 + seed_id)
    cx, cy = 0.5 * width, 0.5 * height
    if rig is None:
        rig = make_scene(0, 200, 20).cam_gt  # only the cameras matter
    n_ref = rig.shape[0]
    Rref = np.stack([rodrigues(c[4:7]) for c in rig])
    fwd = Rref[:, 2, :]  # optical axis of each reference view in world coordinates (R maps world->cam)
    cam_ref = np.zeros((n_query, 15))
    cam_gt = np.zeros((n_query, 15))
    cam_init = np.zeros((n_query, 15))
    uv_ref = np.zeros((n_query * n_match, 2), dtype=np.float32)
    uv_cur = np.zeros((n_query * n_match, 2), dtype=np.float32)
    pan = rng.uniform(n_query, -math.pi, math.pi)
    tilt = np.deg2rad(rng.uniform(n_query, -10.0, 10.0))
    fq = rng.uniform(n_query, 1500.0, 4000.0)
    k1q = rng.uniform(n_query, -0.05, 0.05) if (factor_type & 1) else np.zeros(n_query)
    over = 3  # oversampling factor for candidate pixels
    for q in range(n_query):
        Rq = _rot_x(tilt[q]) @ _rot_y(pan[q])
        ref = int(np.argmax(fwd @ Rq[2, :]))
        cr = rig[ref]
        # sample pixels in the reference image, keep those that land inside the query frame
        cand = n_match * over * 4
        pu = rng.uniform(cand, 8.0, width - 8.0)
        pv = rng.uniform(cand, 8.0, height - 8.0)
        ray_c = np.stack([(pu - cr[2]) / cr[0], (pv - cr[3]) / cr[1], np.ones(cand)], axis=1)
        ray_w = ray_c @ Rref[ref]  # R^T applied: world = R^-1 cam
        pc = ray_w @ Rq.T
        zz = pc[:, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            x = pc[:, 0] / zz
            y = pc[:, 1] / zz
            rad = 1.0 + k1q[q] * (x * x + y * y)
            qu = fq[q] * x * rad + cx
            qv = fq[q] * y * rad + cy
        ok = np.nonzero((zz > 0.1) & (qu >= 8) & (qu <= width - 8) & (qv >= 8) & (qv <= height - 8))[0]
        if len(ok) < n_match:
            # degenerate overlap: fall back to pixels near the reference centre
            ok = np.argsort((pu - cx) ** 2 + (pv - cy) ** 2)[:n_match]
        ok = ok[:n_match]
        s = slice(q * n_match, (q + 1) * n_match)
        uv_ref[s, 0] = (pu[ok] + rng.normal(n_match, noise_px)).astype(np.float32)
        uv_ref[s, 1] = (pv[ok] + rng.normal(n_match, noise_px)).astype(np.float32)
        uv_cur[s, 0] = (qu[ok] + rng.normal(n_match, noise_px)).astype(np.float32)
        uv_cur[s, 1] = (qv[ok] + rng.normal(n_match, noise_px)).astype(np.float32)
        cam_ref[q] = cr
        cam_gt[q] = cr
        cam_gt[q, 0] = cam_gt[q, 1] = fq[q]
        cam_gt[q, 2], cam_gt[q, 3] = cx, cy
        cam_gt[q, 4:7] = rodrigues_inv(Rq)
        cam_gt[q, 10] = k1q[q]
        # init: K = diag(f_ref) with the test image centre, R = R_ref, t = t_ref, dist = dist_ref
        cam_init[q] = cr
        cam_init[q, 1] = cr[0]
        cam_init[q, 2], cam_init[q, 3] = cx, cy
    match_ptr = (np.arange(n_query + 1, dtype=np.int64) * n_match)
    return RelocBatch(n_query=n_query, match_ptr=match_ptr, uv_ref=uv_ref, uv_cur=uv_cur, cam_ref=cam_ref,
                      cam_init=cam_init, cam_gt=cam_gt, factor_type=factor_type)


def _mix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _grid_uniform(seed: int, q: np.ndarray, first: int, count: int) -> np.ndarray:
    """Uniform [0, 1) numbers u[q, i], i = first .. first + count - 1, of a counter-based generator addressed by
    (query, index): a query's numbers do not depend on how many queries are generated or in which chunks."""
    with np.errstate(over="ignore"):
        ctr = q.astype(np.uint64)[:, None] * np.uint64(1 << 16) + np.arange(first + 1, first + count + 1, dtype=np.uint64)[None, :]
        z = _mix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + ctr * _GOLDEN)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def make_reloc_queries(n_query: int, n_match: int = 128, seed_id: int = 0, factor_type: int = 0,
                       rig: np.ndarray | None = None, width: int = 1920, height: int = 1080,
                       noise_px: float = 0.5, chunk: int = 2048) -> RelocBatch:
    """BASELINE C5 at full size (100 000 queries): the workload of make_reloc_batch -- pan / tilt over the covered band,
    f ~ U(1500, 4000), n_match matches against the best-overlap reference view, 0.5 px noise, init = (f_ref, R_ref)
    (run_ptz_reloc.cc:96-104) -- generated a chunk of queries at a time with a counter-based generator, because the
    per-query Python loop of make_reloc_batch takes minutes at this size.  (A different random stream: the two functions do
    not produce the same queries for the same seed.)"""
    seed = SEED_BASE + 0x300000 + seed_id
    cx, cy = 0.5 * width, 0.5 * height
    if rig is None:
        rig = make_scene(0, 200, 20).cam_gt
    Rref = np.stack([rodrigues(c[4:7]) for c in rig])
    fwd = Rref[:, 2, :]
    cand = n_match * 4
    assert 4 + 2 * cand + 8 * n_match < (1 << 16)
    cam_ref = np.zeros((n_query, 15)); cam_gt = np.zeros((n_query, 15)); cam_init = np.zeros((n_query, 15))
    uv_ref = np.zeros((n_query, n_match, 2), dtype=np.float32); uv_cur = np.zeros((n_query, n_match, 2), dtype=np.float32)
    def do_chunk(q0):  # (chunks write disjoint rows of the output arrays: they run on a few threads, numpy releases the GIL in the heavy ops)
        q = np.arange(q0, min(n_query, q0 + chunk))
        nq = len(q)
        head = _grid_uniform(seed, q, 0, 4)
        pan = -math.pi + 2.0 * math.pi * head[:, 0]
        tilt = np.deg2rad(-10.0 + 20.0 * head[:, 1])
        fq = 1500.0 + 2500.0 * head[:, 2]
        k1q = (-0.05 + 0.1 * head[:, 3]) if (factor_type & 1) else np.zeros(nq)
        ct, st_, cp, sp = np.cos(tilt), np.sin(tilt), np.cos(pan), np.sin(pan)
        # Rq = rot_x(tilt) @ rot_y(pan)
        Rq = np.zeros((nq, 3, 3))
        Rq[:, 0, 0] = cp; Rq[:, 0, 2] = sp
        Rq[:, 1, 0] = st_ * sp; Rq[:, 1, 1] = ct; Rq[:, 1, 2] = -st_ * cp
        Rq[:, 2, 0] = -ct * sp; Rq[:, 2, 1] = st_; Rq[:, 2, 2] = ct * cp
        ref = np.argmax(Rq[:, 2, :] @ fwd.T, axis=1)
        cr = rig[ref]                                   # [nq, 15]
        u = _grid_uniform(seed, q, 4, 2 * cand)
        pu = 8.0 + (width - 16.0) * u[:, :cand]
        pv = 8.0 + (height - 16.0) * u[:, cand:]
        ray_c = np.stack([(pu - cr[:, 2:3]) / cr[:, 0:1], (pv - cr[:, 3:4]) / cr[:, 1:2], np.ones_like(pu)], axis=2)  # [nq, cand, 3]
        pc = ray_c @ (Rref[ref] @ np.transpose(Rq, (0, 2, 1)))  # R_ref^T applied to the pixel direction, then R_q
        zz = pc[:, :, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            x = pc[:, :, 0] / zz
            y = pc[:, :, 1] / zz
            rad = 1.0 + k1q[:, None] * (x * x + y * y)
            qu = fq[:, None] * x * rad + cx
            qv = fq[:, None] * y * rad + cy
        ok = (zz > 0.1) & (qu >= 8) & (qu <= width - 8) & (qv >= 8) & (qv <= height - 8)
        first = np.argsort(~ok, axis=1, kind="stable")[:, :n_match]   # the first n_match candidates that land in the query frame
        short = np.flatnonzero(ok.sum(axis=1) < n_match)
        for r in short:  # degenerate overlap: pixels nearest the reference centre, as make_reloc_batch does
            first[r] = np.argsort((pu[r] - cx) ** 2 + (pv[r] - cy) ** 2)[:n_match]
        rows = np.arange(nq)[:, None]
        un = _grid_uniform(seed, q, 4 + 2 * cand, 8 * n_match)
        def gauss(k):  # Box-Muller on columns [2k, 2k + 1) blocks of n_match
            u1 = 1.0 - un[:, (2 * k) * n_match:(2 * k + 1) * n_match]
            u2 = un[:, (2 * k + 1) * n_match:(2 * k + 2) * n_match]
            return noise_px * np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)
        uv_ref[q, :, 0] = (pu[rows, first] + gauss(0)).astype(np.float32)
        uv_ref[q, :, 1] = (pv[rows, first] + gauss(1)).astype(np.float32)
        uv_cur[q, :, 0] = (qu[rows, first] + gauss(2)).astype(np.float32)
        uv_cur[q, :, 1] = (qv[rows, first] + gauss(3)).astype(np.float32)
        cam_ref[q] = cr
        g = cr.copy()
        g[:, 0] = g[:, 1] = fq
        g[:, 2], g[:, 3] = cx, cy
        # rotation vector of Rq = rot_x(tilt) rot_y(pan), per query
        for i in range(nq):
            g[i, 4:7] = rodrigues_inv(Rq[i])
        g[:, 10] = k1q
        cam_gt[q] = g
        ini = cr.copy()
        ini[:, 1] = cr[:, 0]
        ini[:, 2], ini[:, 3] = cx, cy
        cam_init[q] = ini

    starts = list(range(0, n_query, chunk))
    if len(starts) > 4:
        import os
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max(1, min(8, os.cpu_count() or 1))) as pool:
            list(pool.map(do_chunk, starts))
    else:
        for q0 in starts:
            do_chunk(q0)
    match_ptr = np.arange(n_query + 1, dtype=np.int64) * n_match
    return RelocBatch(n_query=n_query, match_ptr=match_ptr, uv_ref=uv_ref.reshape(-1, 2), uv_cur=uv_cur.reshape(-1, 2),
                      cam_ref=cam_ref, cam_init=cam_init, cam_gt=cam_gt, factor_type=factor_type)


_SCENE_ARRAYS = ("obs_uv", "obs_cam", "obs_ray", "ray_weight", "cam_gt", "cam_init", "ray_gt", "ray_init")


def save_scene(path: str, sc: Scene) -> None:
    """A generated scene as a plain .npz (arrays + a JSON header; no pickle), written atomically."""
    import json
    import os
    head = dict(n_cam=sc.n_cam, n_ray=sc.n_ray, width=sc.width, height=sc.height, factor_type=sc.factor_type, seed=sc.seed, meta=sc.meta)
    arrays = {k: getattr(sc, k) for k in _SCENE_ARRAYS}
    if sc.ic_of_cam is not None:
        arrays["ic_of_cam"] = sc.ic_of_cam
    tmp = f"{path}.{os.getpid()}.tmp.npz"
    np.savez(tmp, header=np.frombuffer(json.dumps(head).encode(), dtype=np.uint8), **arrays)
    os.replace(tmp, path)


def load_scene(path: str) -> Scene:
    import json
    with np.load(path, allow_pickle=False) as z:
        head = json.loads(z["header"].tobytes().decode())
        arrays = {k: z[k] for k in _SCENE_ARRAYS}
        ic = z["ic_of_cam"] if "ic_of_cam" in z.files else None
    return Scene(n_cam=head["n_cam"], n_ray=head["n_ray"], width=head["width"], height=head["height"], factor_type=head["factor_type"],
                 seed=head["seed"], meta=head["meta"], ic_of_cam=ic, **arrays)


def _scene_cache_path(cache_dir: str, scene_id: int, n_views: int, obs_per_view: int, kw: dict) -> str:
    import hashlib
    import os
    tag = hashlib.sha256(repr(sorted(kw.items())).encode()).hexdigest()[:10] if kw else "default"
    return os.path.join(cache_dir, f"scene_g{_generator_version()}_{SEED_BASE + scene_id:x}_{n_views}x{obs_per_view}_{tag}.npz")


_GEN_VERSION = None


def _generator_version() -> str:
    """Part of every cache file's name: a hash of this module's source, so that a scene left by another checkout's generator is
    never taken for this one's."""
    global _GEN_VERSION
    if _GEN_VERSION is None:
        import hashlib
        with open(__file__, "rb") as f:
            _GEN_VERSION = hashlib.sha256(f.read()).hexdigest()[:8]
    return _GEN_VERSION


def _make_scene_job(args):
    (scene_id, n_views, obs_per_view), kw, cache_dir = args
    sc = make_scene(scene_id, n_views, obs_per_view, **kw)
    if cache_dir:
        try:
            save_scene(_scene_cache_path(cache_dir, scene_id, n_views, obs_per_view, kw), sc)
        except OSError:
            pass  # the cache is an optimisation only
    return sc


def make_scenes(scene_ids, n_views: int = 200, obs_per_view: int = 500, workers: int | None = None,
                cache_dir: str | None = None, **kw) -> list:
    """make_scene for many ids on several host processes (a C2 scene takes ~0.4 s of numpy; BASELINE C4 wants 1000).
    Call BEFORE the process touches the GPU: the workers are forked.  cache_dir: scenes are deterministic functions of their
    arguments, so generated ones are kept there as .npz files and read back by later calls (bench.py at 1, 2, 4, 8 GPUs on one
    node generates every scene once)."""
    import multiprocessing as mp
    import os
    ids = list(scene_ids)
    if cache_dir:
        os.makedirs(cache_dir, exist_ok=True)
        out = {}
        for i in ids:
            path = _scene_cache_path(cache_dir, i, n_views, obs_per_view, kw)
            if os.path.exists(path):
                try:
                    out[i] = load_scene(path)
                except (OSError, ValueError, KeyError):
                    pass  # unreadable (e.g. another rank is still writing under a different name): generate it
        missing = [i for i in ids if i not in out]
        if missing:
            for i, sc in zip(missing, make_scenes(missing, n_views, obs_per_view, workers, None, _cache_to=cache_dir, **kw)):
                out[i] = sc
        return [out[i] for i in ids]
    cache_to = kw.pop("_cache_to", None)
    if workers is None:
        workers = usable_cores()
        workers = min(workers, 16)
    workers = max(1, min(workers, len(ids)))
    jobs = [((i, n_views, obs_per_view), kw, cache_to) for i in ids]
    if workers == 1:
        return [_make_scene_job(j) for j in jobs]
    with mp.get_context("fork").Pool(workers) as pool:
        return pool.map(_make_scene_job, jobs, chunksize=max(1, len(jobs) // (4 * workers)))


def add_reloc_points(batch: RelocBatch, n_pt: int = 12, noise_px: float = 0.5, depth: float = 60.0) -> RelocBatch:
    """Adds n_pt 2D-3D constraints to every query: world points in front of the ground-truth camera, projected the way
    cv::projectPoints does with the ground-truth camera (its translation is the query's initial one, which the single-view
    solve never changes) and its (k1,k2,p1,p2,k3) reading of the stored distortion."""
    rng = SplitMix64(SEED_BASE + 0x2D3D00 + batch.n_query)
    nq = batch.n_query
    pts2d = np.zeros((nq * n_pt, 2), dtype=np.float32)
    pts3d = np.zeros((nq * n_pt, 3))
    for q in range(nq):
        g = batch.cam_gt[q]
        R = rodrigues(g[4:7])
        t = batch.cam_init[q, 7:10]
        u = rng.uniform(n_pt, 60.0, 2 * g[2] - 60.0)
        v = rng.uniform(n_pt, 60.0, 2 * g[3] - 60.0)
        z = rng.uniform(n_pt, 0.5 * depth, 1.5 * depth)
        # a point at depth z along the (undistorted) pixel direction, then re-projected exactly with distortion
        Pc = np.stack([(u - g[2]) / g[0] * z, (v - g[3]) / g[1] * z, z], axis=1)
        Xw = (Pc - t) @ R          # R^T (P - t)
        P = Xw @ R.T + t
        x, y = P[:, 0] / P[:, 2], P[:, 1] / P[:, 2]
        k1, k2, p1, p2, k3 = g[10], g[11], g[12], g[13], g[14]
        r2 = x * x + y * y
        cd = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2
        xd = x * cd + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * cd + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        sl = slice(q * n_pt, (q + 1) * n_pt)
        pts2d[sl, 0] = (xd * g[0] + g[2] + rng.normal(n_pt, noise_px)).astype(np.float32)
        pts2d[sl, 1] = (yd * g[1] + g[3] + rng.normal(n_pt, noise_px)).astype(np.float32)
        pts3d[sl] = Xw
    batch.point_ptr = np.arange(nq + 1, dtype=np.int64) * n_pt
    batch.pts2d = pts2d
    batch.pts3d = pts3d
    return batch


def add_annotations(scene: Scene, n_annotated: int = 6, pts_per_cam: int = 12, noise_px: float = 0.5,
                    init_rot_sigma_deg: float = 1.0, init_trans_sigma: float = 0.5) -> Scene:
    """Georeferencing stage input (run_ptz_ba.cc:131-155): 2D-3D annotations of world points on the ground plane
    z_w = 0 seen by a few views, the ground-truth T_l_w and a perturbed initial T_l_w (the reference gets its initial
    value from EPnP on the first annotated view, ptzray_optimizer.cc:562-633).  The rig sits 15 m above the ground."""
    rng = SplitMix64(scene.seed ^ 0xA22074)
    Rlw = _rot_x(math.radians(-65.0)) @ _rot_y(math.radians(8.0))   # world (z up) -> local (z forward at pan 0, y down)
    C = np.array([3.0, -45.0, 15.0])                                  # rig centre in world coordinates
    t_lw = -Rlw @ C
    tlw_gt = np.concatenate([rodrigues_inv(Rlw), t_lw])
    step = max(1, scene.n_cam // n_annotated)
    cams = list(range(0, scene.n_cam, step))[:n_annotated]
    uv, xyz, cam = [], [], []
    for ci in cams:
        c = scene.cam_gt[ci]
        R = rodrigues(c[4:7])
        got = 0
        tries = 0
        while got < pts_per_cam and tries < 50 * pts_per_cam:
            tries += 1
            u = float(rng.uniform(1, 40.0, scene.width - 40.0)[0])
            v = float(rng.uniform(1, 40.0, scene.height - 40.0)[0])
            d_l = R.T @ np.array([(u - c[2]) / c[0], (v - c[3]) / c[1], 1.0])
            d_w = Rlw.T @ d_l
            if d_w[2] > -1e-3:
                continue  # does not hit the ground in front of the camera
            lam = -C[2] / d_w[2]
            Xw = C + lam * d_w
            if lam > 400.0:
                continue
            Xl = Rlw @ Xw + t_lw
            P = R @ Xl
            x, y_ = P[0] / P[2], P[1] / P[2]
            r2 = x * x + y_ * y_
            rad = 1.0 + c[10] * r2
            pu = c[0] * x * rad + c[2] + float(rng.normal(1, noise_px)[0])
            pv = c[1] * y_ * rad + c[3] + float(rng.normal(1, noise_px)[0])
            uv.append([pu, pv]); xyz.append(Xw); cam.append(ci)
            got += 1
    scene.obs3d = dict(uv=np.asarray(uv, dtype=np.float32), xyz=np.asarray(xyz, dtype=np.float64), cam=np.asarray(cam, dtype=np.int32))
    scene.tlw_gt = tlw_gt
    pert = rng.normal(3, math.radians(init_rot_sigma_deg))
    scene.tlw_init = np.concatenate([rodrigues_inv(rodrigues(pert) @ Rlw), t_lw + rng.normal(3, init_trans_sigma)])
    return scene


# ------------------------------------------------------------------------------------------------ match table (PTZ-IBA input)
@dataclass
class MatchTable:
    """What LoadImages + LoadMatchesInfo hand to PtzIncrementalOptimizer (data_io.cc:294-400): key points per image and the
    non-empty cells of the N x N MatchesInfo table in table order (row = source image, column = destination image)."""

    n_img: int
    img_wh: np.ndarray      # int32 [n_img, 2]
    kp_ptr: np.ndarray      # int64 [n_img + 1]
    kp_xy: np.ndarray       # float32 [n_kp, 2]
    src: np.ndarray         # int64 [n_pairs]
    dst: np.ndarray         # int64 [n_pairs]
    match_ptr: np.ndarray   # int64 [n_pairs + 1]
    q: np.ndarray           # int32 [n_match]  feature index in the source image
    t: np.ndarray           # int32 [n_match]  feature index in the destination image
    H: np.ndarray           # float64 [n_pairs, 9]  H_j_i (source pixel -> destination pixel), h33 = 1
    h_valid: np.ndarray     # int32 [n_pairs]
    confidence: np.ndarray  # float64 [n_pairs]  min(1, matches / 100) stored as float32 (CalMatchingScore, data_io.cc:358-366)

    @property
    def n_pairs(self) -> int:
        return int(self.src.shape[0])

    def pairs(self):
        return [(int(self.src[p]), int(self.dst[p]),
                 [(int(self.q[k]), int(self.t[k])) for k in range(self.match_ptr[p], self.match_ptr[p + 1])])
                for p in range(self.n_pairs)]


def homography_dlt(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Least-squares homography b ~ H a (normalised DLT) with h33 = 1: stands in for the inlier fit at the end of
    cv::findHomography (data_io.cc:352) on outlier-free synthetic matches."""
    def norm(p):
        c = p.mean(0)
        s = math.sqrt(2.0) / max(np.sqrt(((p - c) ** 2).sum(1)).mean(), 1e-12)
        return np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    Ta, Tb = norm(a), norm(b)
    an = a @ Ta[:2, :2].T + Ta[:2, 2]; bn = b @ Tb[:2, :2].T + Tb[:2, 2]
    A = np.zeros((2 * len(a), 9))
    A[0::2, 0:2] = an; A[0::2, 2] = 1; A[0::2, 6:8] = -bn[:, :1] * an; A[0::2, 8] = -bn[:, 0]
    A[1::2, 3:5] = an; A[1::2, 5] = 1; A[1::2, 6:8] = -bn[:, 1:] * an; A[1::2, 8] = -bn[:, 1]
    h = np.linalg.eigh(A.T @ A)[1][:, 0].reshape(3, 3)  # singular vector of the smallest singular value
    Hm = np.linalg.inv(Tb) @ h @ Ta
    return Hm / Hm[2, 2]


def make_match_table(scene: Scene, min_pair_matches: int = 8, max_pair_gap: int | None = None,
                     bidirectional: bool = True) -> MatchTable:
    """Pairwise matches implied by the tracks of a synthetic scene: every two views of a track are matched, pairs with
    fewer than `min_pair_matches` are dropped, one homography per pair.  bidirectional=False lists every pair once
    (i < j); the reference then can only register images with a HIGHER index than an already registered one, because
    RegisterNextImage reads the table in the (registered source -> new destination) direction only
    (ptz_incremental_optimizer.cc:391).  bidirectional=True also lists (j, i) with the matches swapped and its own fit."""
    n = scene.n_cam
    counts = np.bincount(scene.obs_cam, minlength=n)
    kp_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    order = np.argsort(scene.obs_cam, kind="stable")
    kp_xy = np.ascontiguousarray(scene.obs_uv[order], dtype=np.float32)
    feat_of_obs = np.empty(scene.n_obs, dtype=np.int64)
    feat_of_obs[order] = np.arange(scene.n_obs) - kp_ptr[scene.obs_cam[order]]
    # every (x < y) pair of observations of a track, vectorised over the tracks of equal length
    ray_start = np.concatenate([[0], np.flatnonzero(np.diff(scene.obs_ray)) + 1]).astype(np.int64)
    ray_len = np.diff(np.concatenate([ray_start, [scene.n_obs]]))
    ox_l, oy_l, tid_l = [], [], []
    for L in np.unique(ray_len):
        tr = np.flatnonzero(ray_len == L)
        idx = ray_start[tr][:, None] + np.arange(L)[None, :]
        xs, ys = np.triu_indices(int(L), k=1)
        ox_l.append(idx[:, xs].ravel()); oy_l.append(idx[:, ys].ravel()); tid_l.append(np.repeat(tr, len(xs)))
    ox = np.concatenate(ox_l); oy = np.concatenate(oy_l); tid = np.concatenate(tid_l)
    ci = scene.obs_cam[ox].astype(np.int64); cj = scene.obs_cam[oy].astype(np.int64)
    fi = feat_of_obs[ox]; fj = feat_of_obs[oy]
    if max_pair_gap is not None:
        gap = np.minimum(cj - ci, n - (cj - ci))
        keep = gap <= max_pair_gap
        ci, cj, fi, fj, tid, ox, oy = ci[keep], cj[keep], fi[keep], fj[keep], tid[keep], ox[keep], oy[keep]
    if bidirectional:
        ci, cj, fi, fj, tid = (np.concatenate(p) for p in ((ci, cj), (cj, ci), (fi, fj), (fj, fi), (tid, tid)))
    key = ci * n + cj
    # matches of a pair in track order (the order a per-track loop appends them), pairs in table order
    srt = np.lexsort((fi, tid, key))
    key, ci, cj, fi, fj = key[srt], ci[srt], cj[srt], fi[srt], fj[srt]
    ukey, first, cnt = np.unique(key, return_index=True, return_counts=True)
    good = cnt >= min_pair_matches
    sel = np.repeat(good, cnt)
    ci, cj, fi, fj = ci[sel], cj[sel], fi[sel], fj[sel]
    cnt = cnt[good]
    src = (ukey[good] // n).astype(np.int64); dst = (ukey[good] % n).astype(np.int64)
    match_ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    q = fi.astype(np.int32); t = fj.astype(np.int32)
    H = np.zeros((len(src), 9)); conf = np.zeros(len(src))
    pa = kp_xy[kp_ptr[ci] + fi].astype(np.float64); pb = kp_xy[kp_ptr[cj] + fj].astype(np.float64)
    for p in range(len(src)):
        sl = slice(match_ptr[p], match_ptr[p + 1])
        H[p] = homography_dlt(pa[sl], pb[sl]).reshape(9)
        m = int(cnt[p])
        conf[p] = float(np.float32(1.0) if m >= 100 else np.float32(m) / np.float32(100))
    keys = src
    img_wh = np.tile(np.array([scene.width, scene.height], dtype=np.int32), (n, 1))
    return MatchTable(n, img_wh, kp_ptr, kp_xy, src, dst, match_ptr, q, t, H, np.ones(len(keys), dtype=np.int32), conf)


def make_match_tables(scenes, workers: int | None = None, **kw) -> list:
    """make_match_table for many scenes on several host processes (1 s of numpy per C2 scene).  Call BEFORE the process touches
    the GPU: the workers are forked."""
    import multiprocessing as mp
    import os
    scenes = list(scenes)
    if workers is None:
        workers = usable_cores()
        workers = min(workers, 16)
    workers = max(1, min(workers, len(scenes)))
    if workers == 1:
        return [make_match_table(sc, **kw) for sc in scenes]
    global _MT_SCENES, _MT_KW
    _MT_SCENES, _MT_KW = scenes, kw  # inherited by the forked workers: the scenes need not be pickled
    try:
        with mp.get_context("fork").Pool(workers) as pool:
            return pool.map(_match_table_job, range(len(scenes)), chunksize=1)
    finally:
        _MT_SCENES, _MT_KW = None, None


_MT_SCENES, _MT_KW = None, None


def _match_table_job(i):
    return make_match_table(_MT_SCENES[i], **_MT_KW)
