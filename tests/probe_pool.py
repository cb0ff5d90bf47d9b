import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as ge
import host_util as hu
pkg = ge.load_package(); orc = ge.load_oracle()
import incremental_oracle as io
sc = pkg.synth.make_scene(3, 24, 100)
tb = pkg.synth.make_match_table(sc, bidirectional=False)
cam0 = np.zeros((tb.n_img, 15)); cam0[:, 0] = cam0[:, 1] = 1.0
o = io.IncrementalOracle(tb, cam0, 200, jacobian_mode=orc.JAC_NUMERIC); o.solve()
for rep in range(int(os.environ.get('REPS', 4))):
    if rep == 2:
        pkg.api.trim_cache()
    if rep == 3:
        # dirty the pool with a different workload first
        pkg.api.ba_solve(pkg.synth.make_scene(9, 30, 120))
    ok, cam, reg, events, nit = hu.incremental_solve(tb, cam0, max_iter=200)
    same = events == o.events
    print("rep", rep, "equal", same, "nit", nit, o.lm_iterations, "reg", len(reg), len(o.reg))
    if not same:
        for k, (a, b) in enumerate(zip(events, o.events)):
            if a != b:
                print("  first diff at", k, a, b); break
