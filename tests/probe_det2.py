import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as ge
import host_util as hu
pkg = ge.load_package()
rb = pkg.synth.make_reloc_batch(37, 128, seed_id=4)
base = pkg.api.krt_solve_batch(rb)
sc = pkg.synth.make_scene(9, 30, 120)
for rep in range(6):
    if rep % 2: pkg.api.ba_solve(sc)
    cam, summ, acc, _ = pkg.api.krt_solve_batch(rb)
    print("krt rep", rep, np.array_equal(cam, base[0]), summ == base[1], np.array_equal(acc, base[2]))
# orchestration repeated: compare the C++ runs with each other, event by event and camera by camera
sc2 = pkg.synth.make_scene(3, 24, 100)
tb = pkg.synth.make_match_table(sc2, bidirectional=False)
cam0 = np.zeros((tb.n_img, 15)); cam0[:, 0] = cam0[:, 1] = 1.0
runs = [hu.incremental_solve(tb, cam0, max_iter=200) for _ in range(4)]
for r in runs[1:]:
    print("orchestration equal to first run: events", r[3] == runs[0][3], "cams", np.array_equal(r[1], runs[0][1]), "nit", r[4], runs[0][4])
