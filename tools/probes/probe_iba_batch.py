"""Probe (not a test): N rigs through the incremental pipeline in lock step (PtzIncrementalOptimizer::SolveBatch) against one rig alone."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
views = int(sys.argv[2]) if len(sys.argv) > 2 else 200
scenes = pkg.synth.make_scenes(range(n), views, 500, cache_dir="/tmp/ptz_scene_cache")
t = time.perf_counter(); tables = pkg.synth.make_match_tables(scenes); print("match tables %.1f s" % (time.perf_counter() - t), flush=True)
cam0 = []
for tb in tables:
    c = np.zeros((tb.n_img, 15)); c[:, 0] = c[:, 1] = 1.0
    cam0.append(c)
pkg.hostlib.incremental_solve(tables[0], cam0[0])
t = time.perf_counter(); r1 = pkg.hostlib.incremental_solve(tables[0], cam0[0]); d = time.perf_counter() - t
print("solo: %.1f ms, %.0f views/s" % (1e3 * d, len(r1["registered"]) / d), r1["timing_ms"], flush=True)
pkg.hostlib.incremental_solve_batch(tables[:4], cam0[:4])
for rep in range(2):
    t = time.perf_counter(); res, st = pkg.hostlib.incremental_solve_batch(tables, cam0, events_as_array=True); d = time.perf_counter() - t
    reg = sum(len(r["registered"]) for r in res)
    print(json.dumps(dict(rigs=n, wall_total_ms=1e3 * d, views_per_s=reg / d, **st)), flush=True)
assert [tuple(int(x) for x in row) for row in res[0]["events"]] == r1["events"] and np.array_equal(res[0]["cameras"], r1["cameras"])
if os.environ.get("PTZ_PROBE_RIG_TIMING"):
    keys = sorted(res[0]["timing_ms"].keys())
    print("per-rig host timing, mean over rigs (ms):", {k: round(float(np.mean([r["timing_ms"][k] for r in res])), 2) for k in keys})
    print("  max over rigs:", {k: round(float(np.max([r["timing_ms"][k] for r in res])), 2) for k in keys})
