"""Probe: PTZ-IBA on one 200-view rig alone, wall and split, a few repetitions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
sc = pkg.synth.make_scene(0, 200, 500)
tb = pkg.synth.make_match_table(sc)
cam0 = np.zeros((tb.n_img, 15)); cam0[:, 0] = cam0[:, 1] = 1.0
pkg.hostlib.incremental_solve(tb, cam0)
for _ in range(4):
    t = time.perf_counter(); r = pkg.hostlib.incremental_solve(tb, cam0); d = time.perf_counter() - t
    print("solo: %.1f ms, %.0f views/s" % (1e3 * d, len(r["registered"]) / d), {k: round(float(v), 1) for k, v in r["timing_ms"].items()}, flush=True)
