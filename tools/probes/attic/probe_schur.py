import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
B = int(sys.argv[1])
base = [pkg.synth.make_scene(s, 200, 500) for s in range(4)]
b = pkg.api.BaBatch([base[i % 4] for i in range(B)], max_num_iterations=6); b.set_state(); b.solve()
b.set_profiling(True); b.solve(); p = b.get_profile()
print(os.environ.get("PTZCALIB_LIB", "product")[-22:], {k: round(v["ms"] / max(v["launches"], 1), 3) for k, v in p.items() if k in ("schur", "rhs", "linearize", "eval", "chol_syrk")})
