"""Plain workload for the rocprofv3 PMC passes: N distinct C2-shaped scenes (seeds 0..N-1, bench.py's C4 workload), ONE solve.
The generated scenes are cached in /tmp so that the PMC passes of one box do not each spend a minute generating them."""
import os, pickle, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cache = f"/tmp/ptz_c4_scenes_{N}.pkl"
if os.path.exists(cache):
    scenes = pickle.load(open(cache, "rb"))
else:
    scenes = pkg.synth.make_scenes(range(N), 200, 500)
    pickle.dump(scenes, open(cache, "wb"), protocol=4)
b = pkg.api.BaBatch(scenes); b.set_state()
s = b.solve()
print("lm_steps", sum(x["num_lm_steps"] for x in s), "ms", b.last_solve_ms())
