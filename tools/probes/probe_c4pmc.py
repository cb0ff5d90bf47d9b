"""Plain workload for the rocprofv3 PMC passes: N distinct C2-shaped scenes (seeds 0..N-1, bench.py's C4 workload), ONE solve.
The generated scenes are cached (synth's .npz scene cache, no pickle) so that the PMC passes of one box do not each spend a minute
generating them.  The first, unprofiled run of probe_profile2.sh fills the cache before any profiler preload touches the GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
scenes = pkg.synth.make_scenes(range(N), 200, 500, cache_dir=os.environ.get("PTZ_SCENE_CACHE", "/tmp/ptz_scene_cache"))
b = pkg.api.BaBatch(scenes); b.set_state()
s = b.solve()
print("lm_steps", sum(x["num_lm_steps"] for x in s), "ms", b.last_solve_ms())
