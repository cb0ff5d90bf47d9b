"""ctypes access to ptz-calib_amd/libptzcalib_host.so (the C++ PTZRayOptimizer / KRTOptimizer mirror) for the tests."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.environ.get("PTZCALIB_HOST_LIB", os.path.join(ROOT, "ptz-calib_amd", "libptzcalib_host.so")))
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def tracks_build(pairs, min_len=4):
    n = len(pairs)
    src = np.array([p[0] for p in pairs], dtype=np.int64); dst = np.array([p[1] for p in pairs], dtype=np.int64)
    ptr = np.concatenate([[0], np.cumsum([len(p[2]) for p in pairs])]).astype(np.int64)
    q = np.array([m[0] for p in pairs for m in p[2]], dtype=np.int32); t = np.array([m[1] for p in pairs for m in p[2]], dtype=np.int32)
    tid = C.POINTER(C.c_int32)(); tptr = C.POINTER(C.c_int64)(); ei = C.POINTER(C.c_int32)(); ef = C.POINTER(C.c_int32)()
    nt = lib().ptzh_tracks_build(n, _p(src), _p(dst), _p(ptr), _p(q), _p(t), min_len, C.byref(tid), C.byref(tptr), C.byref(ei), C.byref(ef))
    assert nt != -2, "TracksBuilder::ExportFlat differs from ExportToSTL"
    out = {int(tid[k]): {int(ei[e]): int(ef[e]) for e in range(tptr[k], tptr[k + 1])} for k in range(nt)}
    for x in (tid, tptr, ei, ef):
        lib().ptzh_free(x)
    return out


def scene_to_features_matches(sc, n_img=None):
    """Turn a packed synthetic scene into (keypoints per image, pairwise match lists): consecutive views of every
    track are matched, which is enough for the union-find to rebuild the same tracks."""
    n_img = sc.n_cam if n_img is None else n_img
    kps = [[] for _ in range(n_img)]
    feat_of_obs = np.zeros(sc.n_obs, dtype=np.int64)
    for a in range(sc.n_obs):
        c = int(sc.obs_cam[a])
        feat_of_obs[a] = len(kps[c])
        kps[c].append(sc.obs_uv[a])
    pairs = {}
    start = 0
    for a in range(1, sc.n_obs + 1):
        if a == sc.n_obs or sc.obs_ray[a] != sc.obs_ray[start]:
            for k in range(start, a - 1):
                i, j = int(sc.obs_cam[k]), int(sc.obs_cam[k + 1])
                pairs.setdefault((i, j), []).append((int(feat_of_obs[k]), int(feat_of_obs[k + 1])))
            start = a
    plist = [(i, j, m) for (i, j), m in sorted(pairs.items())]
    return kps, plist


def epnp(xyz, uv, K, dist):
    """cv::solvePnP(EPNP) replacement of the host library: returns (ok, R 3x3, t 3)."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float64); uv = np.ascontiguousarray(uv, dtype=np.float32)
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9); dist = np.ascontiguousarray(dist, dtype=np.float64)
    R = np.zeros(9); t = np.zeros(3)
    ok = lib().ptzh_epnp(len(xyz), _p(xyz), _p(uv), _p(K), _p(dist), _p(R), _p(t))
    return bool(ok), R.reshape(3, 3), t


def ptzray_solve(kps, plist, cam15, cand_ids=(), max_iter=200, ftype=0, on_device=True, annotations=None):
    """annotations: None or (cam index array, uv [n,2] f32, xyz [n,3] f64) sorted by camera.  With annotations the
    returned dict `packed` also holds tlw_init / tlw / tlw_ok."""
    from ctypes import POINTER, byref, c_double, c_float, c_int32, c_int64
    import __graft_entry__ as ge
    api = ge.load_package().api
    n_img = len(kps)
    kp_ptr = np.concatenate([[0], np.cumsum([len(k) for k in kps])]).astype(np.int64)
    kp_xy = np.ascontiguousarray(np.concatenate([np.asarray(k, dtype=np.float32).reshape(-1, 2) for k in kps]), dtype=np.float32)
    src = np.array([p[0] for p in plist], dtype=np.int64); dst = np.array([p[1] for p in plist], dtype=np.int64)
    mptr = np.concatenate([[0], np.cumsum([len(p[2]) for p in plist])]).astype(np.int64)
    q = np.array([m[0] for p in plist for m in p[2]], dtype=np.int32); t = np.array([m[1] for p in plist for m in p[2]], dtype=np.int32)
    cam = np.array(cam15, dtype=np.float64, order="C").copy()
    cand = np.array(list(cand_ids), dtype=np.int64)
    errors = np.zeros(3)
    summ = api.LmSummary()
    n_obs = c_int32(); n_ray = c_int32()
    puv = POINTER(c_float)(); pcam = POINTER(c_int32)(); pray = POINTER(c_int32)(); pw = POINTER(c_double)()
    pc15 = POINTER(c_double)(); pr3 = POINTER(c_double)(); pci = POINTER(c_int64)()
    ann_ptr = ann_uv = ann_xyz = None
    if annotations is not None:
        acam, auv, axyz = annotations
        acam = np.asarray(acam)
        assert np.all(np.diff(acam) >= 0)
        ann_ptr = np.searchsorted(acam, np.arange(n_img + 1)).astype(np.int64)
        ann_uv = np.ascontiguousarray(auv, dtype=np.float32); ann_xyz = np.ascontiguousarray(axyz, dtype=np.float64)
    tlw_out = np.zeros(14)
    ok = lib().ptzh_ptzray_georef(n_img, _p(kp_ptr), _p(kp_xy), len(plist), _p(src), _p(dst), _p(mptr), _p(q), _p(t), _p(cam),
                                  _p(ann_ptr), _p(ann_uv), _p(ann_xyz),
                                  _p(cand) if len(cand) else None, len(cand), max_iter, ftype, int(on_device), _p(errors),
                                  byref(summ), _p(tlw_out), byref(n_obs), byref(n_ray), byref(puv), byref(pcam), byref(pray),
                                  byref(pw), byref(pc15), byref(pr3), byref(pci))
    no, nr = n_obs.value, n_ray.value
    ncam = len(cand) if len(cand) else n_img
    packed = dict(obs_uv=np.ctypeslib.as_array(puv, (no, 2)).copy(), obs_cam=np.ctypeslib.as_array(pcam, (no,)).copy(),
                  obs_ray=np.ctypeslib.as_array(pray, (no,)).copy(), ray_weight=np.ctypeslib.as_array(pw, (nr,)).copy(),
                  cam=np.ctypeslib.as_array(pc15, (ncam, 15)).copy(), ray=np.ctypeslib.as_array(pr3, (nr, 3)).copy(),
                  cam_image=np.ctypeslib.as_array(pci, (ncam,)).copy(),
                  tlw_init=tlw_out[:6].copy(), tlw=tlw_out[6:12].copy(), tlw_ok=bool(tlw_out[12]), n_obs3d=int(tlw_out[13]))
    for x in (puv, pcam, pray, pw, pc15, pr3, pci):
        lib().ptzh_free(x)
    return bool(ok), cam, errors, summ.as_dict(), packed


def krt_solve(cam_ref, cam_cur, uv_ref, uv_cur, max_iter=200, max_err=100.0, ftype=0):
    import __graft_entry__ as ge
    api = ge.load_package().api
    cur = np.array(cam_cur, dtype=np.float64).copy()
    ref = np.ascontiguousarray(cam_ref, dtype=np.float64)
    ur = np.ascontiguousarray(uv_ref, dtype=np.float32); uc = np.ascontiguousarray(uv_cur, dtype=np.float32)
    nit = C.c_int32(); summ = api.LmSummary()
    ok = lib().ptzh_krt_solve(_p(ref), _p(cur), len(ur), _p(ur), _p(uc), max_iter, C.c_double(max_err), ftype, C.byref(nit), C.byref(summ))
    return bool(ok), cur, nit.value, summ.as_dict()


def krt_solve_2d3d(cam_ref, cam_cur, uv_ref, uv_cur, pts2d, pts3d, max_iter=200, max_err=100.0, ftype=0, swapped=False):
    """KRTOptimizer with Add2d2dConstraints + Add2d3dConstraints.  Returns (code, cam, num_iter_, summary, [err2d2d, err2d3d]);
    code 1 = solved, 0 = Solve returned false, -2 = Add2d3dConstraints before Add2d2dConstraints threw."""
    import __graft_entry__ as ge
    api = ge.load_package().api
    cur = np.array(cam_cur, dtype=np.float64).copy()
    ref = np.ascontiguousarray(cam_ref, dtype=np.float64)
    ur = np.ascontiguousarray(uv_ref, dtype=np.float32); uc = np.ascontiguousarray(uv_cur, dtype=np.float32)
    p2 = np.ascontiguousarray(pts2d, dtype=np.float32); p3 = np.ascontiguousarray(pts3d, dtype=np.float64)
    nit = C.c_int32(); summ = api.LmSummary(); err = np.zeros(2)
    code = lib().ptzh_krt_solve_2d3d(_p(ref), _p(cur), len(ur), _p(ur), _p(uc), len(p2), _p(p2), _p(p3), max_iter,
                                     C.c_double(max_err), ftype, int(swapped), C.byref(nit), C.byref(summ), _p(err))
    return int(code), cur, nit.value, summ.as_dict(), err


def incremental_solve(table, cam15, max_iter=200, seeds=()):
    incremental_solve.timing = np.zeros(7)
    """PtzIncrementalOptimizer::Solve through the C++ class.  Returns (ok, cam15, registered ids, events, lm_iterations)."""
    tb = table
    cam = np.array(cam15, dtype=np.float64, order="C").copy()
    reg = np.zeros(tb.n_img, dtype=np.int32)
    max_events = 16 * tb.n_img + 256
    ev = np.zeros((max_events, 4), dtype=np.int64)
    seeds = np.array(list(seeds), dtype=np.int64)
    nit = C.c_int64(0); solved = C.c_int32(0)
    H = np.ascontiguousarray(tb.H, dtype=np.float64); hv = np.ascontiguousarray(tb.h_valid, dtype=np.int32)
    conf = np.ascontiguousarray(tb.confidence, dtype=np.float64)
    wh = np.ascontiguousarray(tb.img_wh, dtype=np.int32)
    ne = lib().ptzh_incremental_solve(tb.n_img, _p(tb.kp_ptr), _p(tb.kp_xy), _p(wh), tb.n_pairs, _p(tb.src), _p(tb.dst),
                                      _p(tb.match_ptr), _p(tb.q), _p(tb.t), _p(H), _p(hv), _p(conf), _p(cam),
                                      _p(seeds) if len(seeds) else None, len(seeds), max_iter, _p(reg), _p(ev), max_events,
                                      C.byref(nit), C.byref(solved), _p(incremental_solve.timing))
    ok = bool(solved.value)
    events = [tuple(int(x) for x in row) for row in ev[:max(ne, 0)]]
    return ok, cam, sorted(int(i) for i in np.flatnonzero(reg)), events, int(nit.value)


def ptzray_solve_shared(kps, plist, cam15, shared_ic_ids, max_iter=200, ftype=0):
    """PTZRayOptimizer with SetSharedIntrinsics(shared_ic_ids) through the C++ class.  Returns (ok, cam15, summary)."""
    import __graft_entry__ as ge
    api = ge.load_package().api
    n_img = len(kps)
    kp_ptr = np.concatenate([[0], np.cumsum([len(k) for k in kps])]).astype(np.int64)
    kp_xy = np.ascontiguousarray(np.concatenate([np.asarray(k, dtype=np.float32).reshape(-1, 2) for k in kps]), dtype=np.float32)
    src = np.array([p[0] for p in plist], dtype=np.int64); dst = np.array([p[1] for p in plist], dtype=np.int64)
    mptr = np.concatenate([[0], np.cumsum([len(p[2]) for p in plist])]).astype(np.int64)
    q = np.array([m[0] for p in plist for m in p[2]], dtype=np.int32); t = np.array([m[1] for p in plist for m in p[2]], dtype=np.int32)
    cam = np.array(cam15, dtype=np.float64, order="C").copy()
    ids = np.ascontiguousarray(shared_ic_ids, dtype=np.int64)
    summ = api.LmSummary()
    ok = lib().ptzh_ptzray_solve_shared(n_img, _p(kp_ptr), _p(kp_xy), len(plist), _p(src), _p(dst), _p(mptr), _p(q), _p(t), _p(cam),
                                        _p(ids), max_iter, ftype, C.byref(summ))
    return bool(ok), cam, summ.as_dict()
