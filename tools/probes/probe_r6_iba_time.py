import os, sys, time, json
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ["GRAFT_REPO_ROOT"])
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
import ctypes as C
n, views = 64, 200
scenes = pkg.synth.make_scenes(range(n), views, 500, cache_dir="/tmp/ptz_scene_cache")
tables = pkg.synth.make_match_tables(scenes)
cam0 = []
for tb in tables:
    c = np.zeros((tb.n_img, 15)); c[:, 0] = c[:, 1] = 1.0
    cam0.append(c)
hl = pkg.hostlib
res, st = hl.incremental_solve_batch(tables[:4], cam0[:4])
# re-implement with timers
from concurrent.futures import ThreadPoolExecutor
L = hl.lib(); _p = hl._p
L.ptzh_inc_create.restype = C.c_void_p
def create(args):
    tb, cam15 = args
    cam = np.array(cam15, dtype=np.float64, order="C").copy()
    H = np.ascontiguousarray(tb.H, dtype=np.float64); hv = np.ascontiguousarray(tb.h_valid, dtype=np.int32)
    conf = np.ascontiguousarray(tb.confidence, dtype=np.float64); wh = np.ascontiguousarray(tb.img_wh, dtype=np.int32)
    h = L.ptzh_inc_create(tb.n_img, _p(tb.kp_ptr), _p(tb.kp_xy), _p(wh), tb.n_pairs, _p(tb.src), _p(tb.dst), _p(tb.match_ptr), _p(tb.q), _p(tb.t), _p(H), _p(hv), _p(conf), _p(cam), None, 0, 200)
    return C.c_void_p(h)
for rep in range(2):
    t0 = time.perf_counter()
    with ThreadPoolExecutor(16) as pool:
        handles = list(pool.map(create, zip(tables, cam0)))
        t1 = time.perf_counter()
        arr = (C.c_void_p * len(handles))(*handles)
        stats = np.zeros(8)
        L.ptzh_inc_solve_batch(arr, len(handles), 0, _p(stats))
        t2 = time.perf_counter()
        for tb, h in zip(tables, handles):
            L.ptzh_inc_destroy(h)
        t3 = time.perf_counter()
    print("create %.1f ms, solve %.1f ms (in-call wall_ms %.1f), destroy %.1f ms" % (1e3*(t1-t0), 1e3*(t2-t1), stats[7], 1e3*(t3-t2)))
