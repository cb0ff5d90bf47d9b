"""Probe (not a test): the 1000-distinct-seed C4 batch, per-scene accuracy.  Prints the scenes whose focal lengths do not come
back at noise level, with their LM bookkeeping, and writes every scene's (iterations, final cost, mean focal error) to
gpurun_out/<tag>_<label>.json so that two builds / kernels can be compared scene by scene.   usage: probe_c4_check.py <label> [n]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
label = sys.argv[1] if len(sys.argv) > 1 else "run"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
scenes = pkg.synth.make_scenes(range(n), 200, 500, cache_dir="/tmp/ptz_scene_cache")
b = pkg.api.BaBatch(scenes); b.set_state()
t = time.perf_counter(); summ = b.solve(); dt = time.perf_counter() - t
cams, rays = b.get_state(); b.close()
err = np.array([np.abs(cams[i][:, 0] - scenes[i].cam_gt[:, 0]).mean() for i in range(n)])
rows = [dict(i=i, it=summ[i]["num_iterations"], steps=summ[i]["num_lm_steps"], term=summ[i]["termination_type"],
             cost=summ[i]["final_cost"], cost0=summ[i]["initial_cost"], ferr=float(err[i])) for i in range(n)]
bad = [r for r in rows if r["ferr"] > 2.5 or r["term"] != 0]
print(label, "solve %.1f ms, lm steps %d, bad scenes %d:" % (1e3 * dt, sum(r["steps"] for r in rows), len(bad)))
for r in bad[:20]:
    print("  ", r)
os.makedirs(os.path.join(ROOT, "gpurun_out", "c4check"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "c4check", label + ".json"), "w"))
