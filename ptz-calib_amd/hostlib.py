"""ctypes access to libptzcalib_host.so, the C++ mirror of the reference's optimizer classes (ptz-calib_amd/host/):
only what bench.py needs -- the PTZ-IBA orchestration (PtzIncrementalOptimizer).  Plumbing, not the product."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libptzcalib_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python __graft_entry__.py` (or make -C ptz-calib_amd/host)")
        _lib = C.CDLL(path)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def incremental_solve(table, cam15, max_iter: int = 200, seeds=()):
    """PtzIncrementalOptimizer(features, matches_info, cameras, max_iter).Solve(cameras, reg_image_ids)
    (src/core/ptz_incremental_optimizer.h:21-44).  `table` is a synth.MatchTable.  Returns a dict."""
    tb = table
    cam = np.array(cam15, dtype=np.float64, order="C").copy()
    reg = np.zeros(tb.n_img, dtype=np.int32)
    max_events = 16 * tb.n_img + 256
    ev = np.zeros((max_events, 4), dtype=np.int64)
    seeds = np.array(list(seeds), dtype=np.int64)
    nit = C.c_int64(0)
    solved = C.c_int32(0)
    timing = np.zeros(7)
    H = np.ascontiguousarray(tb.H, dtype=np.float64)
    hv = np.ascontiguousarray(tb.h_valid, dtype=np.int32)
    conf = np.ascontiguousarray(tb.confidence, dtype=np.float64)
    wh = np.ascontiguousarray(tb.img_wh, dtype=np.int32)
    ne = lib().ptzh_incremental_solve(tb.n_img, _p(tb.kp_ptr), _p(tb.kp_xy), _p(wh), tb.n_pairs, _p(tb.src), _p(tb.dst),
                                      _p(tb.match_ptr), _p(tb.q), _p(tb.t), _p(H), _p(hv), _p(conf), _p(cam),
                                      _p(seeds) if len(seeds) else None, len(seeds), max_iter, _p(reg), _p(ev), max_events,
                                      C.byref(nit), C.byref(solved), _p(timing))
    events = [tuple(int(x) for x in row) for row in ev[:max(ne, 0)]]
    return dict(ok=bool(solved.value), cameras=cam, registered=sorted(int(i) for i in np.flatnonzero(reg)), events=events,
                lm_iterations=int(nit.value),
                timing_ms=dict(ranking=timing[0], bundle_total=timing[1], bundle_device=timing[2], registration_total=timing[3],
                               registration_device=timing[4], construct=timing[5], solve=timing[6]))


def incremental_solve_batch(tables, cam15s, max_iter: int = 200, device_id: int = 0, events_as_array: bool = False):
    """N rigs in lock step on one GPU (PtzIncrementalOptimizer::SolveBatch, host/device_batcher.h): every rig's optimizer on a
    host thread of its own, the bundle adjustments and registration attempts of each round batched into one library call each.
    Returns (list of per-rig dicts as incremental_solve gives them, without timing; batch statistics)."""
    from concurrent.futures import ThreadPoolExecutor
    L = lib()
    L.ptzh_inc_create.restype = C.c_void_p

    def create(args):  # marshals one rig's tables into the C++ structures (ctypes releases the GIL: the rigs go in parallel)
        tb, cam15 = args
        cam = np.array(cam15, dtype=np.float64, order="C").copy()
        H = np.ascontiguousarray(tb.H, dtype=np.float64)
        hv = np.ascontiguousarray(tb.h_valid, dtype=np.int32)
        conf = np.ascontiguousarray(tb.confidence, dtype=np.float64)
        wh = np.ascontiguousarray(tb.img_wh, dtype=np.int32)
        h = L.ptzh_inc_create(tb.n_img, _p(tb.kp_ptr), _p(tb.kp_xy), _p(wh), tb.n_pairs, _p(tb.src), _p(tb.dst), _p(tb.match_ptr), _p(tb.q),
                              _p(tb.t), _p(H), _p(hv), _p(conf), _p(cam), None, 0, max_iter)
        return C.c_void_p(h)

    def result(args):
        tb, h = args
        cam = np.zeros((tb.n_img, 15))
        reg = np.zeros(tb.n_img, dtype=np.int32)
        max_events = 16 * tb.n_img + 256
        ev = np.zeros((max_events, 4), dtype=np.int64)
        nit = C.c_int64(0)
        solved = C.c_int32(0)
        ne = L.ptzh_inc_result(h, _p(cam), _p(reg), _p(ev), max_events, C.byref(nit), C.byref(solved))
        L.ptzh_inc_destroy(h)
        # (events_as_array: the int64 [n, 4] rows themselves -- turning ~3000 rows per rig into Python tuples is a third of this
        #  function's time outside the library call for 64 rigs, all of it under the interpreter lock)
        events = ev[:max(ne, 0)].copy() if events_as_array else [tuple(int(x) for x in row) for row in ev[:max(ne, 0)]]
        return dict(ok=bool(solved.value), cameras=cam, registered=np.flatnonzero(reg).tolist(),
                    events=events, lm_iterations=int(nit.value))

    workers = max(1, min(int(os.environ.get("PTZ_HOST_MARSHAL_THREADS", "16")), len(tables), os.cpu_count() or 1))  # (marshalling and tear-down of the rigs; measured, 64 rigs: 16 threads 383-430 ms for the whole call, 64 threads 547-598 -- they queue for the interpreter lock)
    with ThreadPoolExecutor(workers) as pool:
        handles = list(pool.map(create, zip(tables, cam15s)))
        arr = (C.c_void_p * len(handles))(*handles)
        stats = np.zeros(8)
        L.ptzh_inc_solve_batch(arr, len(handles), device_id, _p(stats))
        out = list(pool.map(result, zip(tables, handles)))
    return out, dict(rounds=int(stats[0]), ba_batches=int(stats[1]), ba_problems=int(stats[2]), krt_launches=int(stats[3]),
                     krt_queries=int(stats[4]), ba_ms=float(stats[5]), krt_ms=float(stats[6]), wall_ms=float(stats[7]))
