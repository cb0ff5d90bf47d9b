import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
sc = pkg.synth.make_scene(1, 200, 500)
tb = pkg.synth.make_match_table(sc)
cam0 = np.zeros((tb.n_img, 15)); cam0[:, 0] = cam0[:, 1] = 1.0
pkg.hostlib.incremental_solve(tb, cam0)
os.environ["PTZ_BA_DEBUG_TIMING"] = "1"
t = time.time(); r = pkg.hostlib.incremental_solve(tb, cam0); print("wall", time.time() - t, r["timing_ms"], file=sys.stderr)
