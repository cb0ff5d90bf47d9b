"""Probe (not a test): the full incremental pipeline on one 200-view rig (bench.py's iba leg), three timed runs."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
sc = pkg.synth.make_scene(0, 200, 500)
tb = pkg.synth.make_match_table(sc)
cam0 = np.zeros((tb.n_img, 15)); cam0[:, 0] = cam0[:, 1] = 1.0
pkg.hostlib.incremental_solve(tb, cam0, max_iter=200)
for _ in range(3):
    t = time.perf_counter(); r = pkg.hostlib.incremental_solve(tb, cam0, max_iter=200); d = time.perf_counter() - t
    print(json.dumps({"wall_ms": round(1e3 * d, 2), "views_per_s": round(len(r["registered"]) / d), "lm_iterations": r["lm_iterations"],
                      "timing_ms": {k: round(float(v), 2) for k, v in r["timing_ms"].items()}}))
