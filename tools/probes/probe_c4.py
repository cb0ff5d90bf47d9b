"""Probe (not a test): the C4 batch (1000 distinct seeds) -- active scenes per pass, launch shapes used, time per solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
scenes = pkg.synth.make_scenes(range(n), 200, 500)
b = pkg.api.BaBatch(scenes); b.set_state(); summ = b.solve()
its = np.array([s["num_lm_steps"] for s in summ])
print("lm steps: mean %.1f max %d; active scenes after pass p:" % (its.mean(), its.max()), [int((its > p).sum()) for p in (0, 5, 10, 15, 20, 25, 30, 40, 50, 75, 100)])
for _ in range(2):
    t = time.perf_counter(); b.solve(); print("solve %.1f ms (device %.1f)" % (1e3 * (time.perf_counter() - t), b.last_solve_ms()))
os.environ["PTZ_BA_DEBUG_TIMING"] = "1"
b.solve()
