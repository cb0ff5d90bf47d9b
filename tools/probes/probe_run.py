"""Plain workload for rocprofv3: B scenes, R solves, no in-library event profiling."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1
base = [pkg.synth.make_scene(s, 200, 500) for s in range(min(B, 4))]
b = pkg.api.BaBatch([base[i % len(base)] for i in range(B)]); b.set_state()
for _ in range(R):
    s = b.solve()
print("lm_steps", sum(x["num_lm_steps"] for x in s), "ms", b.last_solve_ms())
