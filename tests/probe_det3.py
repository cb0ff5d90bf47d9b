import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as ge
import host_util as hu
pkg = ge.load_package()
sc2 = pkg.synth.make_scene(3, 24, 100)
tb = pkg.synth.make_match_table(sc2, bidirectional=False)
cam0 = np.zeros((tb.n_img, 15)); cam0[:, 0] = cam0[:, 1] = 1.0
for r in range(2):
    sys.stderr.write("=== run %d\n" % r); sys.stderr.flush()
    hu.incremental_solve(tb, cam0, max_iter=200)
