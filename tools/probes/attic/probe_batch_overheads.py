"""Probe (not a test): what a batched bundle adjustment of the lock-step PTZ-IBA costs besides the device solve -- batch creation
(host structure + uploads), set_state, solve, get_state, destroy -- for 64 rigs at several model sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for views, opv in ((20, 500), (50, 500), (100, 500), (200, 500)):
    scenes = pkg.synth.make_scenes(range(n), views, opv, cache_dir="/tmp/ptz_scene_cache")
    for rep in range(2):
        t0 = time.perf_counter(); b = pkg.api.BaBatch(scenes); t1 = time.perf_counter()
        b.set_state(); t2 = time.perf_counter()
        s = b.solve(); t3 = time.perf_counter()
        b.get_state(); t4 = time.perf_counter()
        b.close(); t5 = time.perf_counter()
    print(f"{n} rigs x {views} views: create {1e3*(t1-t0):.1f} ms, set_state {1e3*(t2-t1):.1f}, solve {1e3*(t3-t2):.1f} (device {b.last_ms if hasattr(b,'last_ms') else 0:.1f}), "
          f"get_state {1e3*(t4-t3):.1f}, destroy {1e3*(t5-t4):.1f}; lm steps {sum(x['num_lm_steps'] for x in s)}", flush=True)
os.environ["PTZ_BA_DEBUG_TIMING"] = "1"
b = pkg.api.BaBatch(scenes); b.set_state(); b.solve(); b.close()
