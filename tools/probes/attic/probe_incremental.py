"""Probe: wall time of the PTZ-IBA orchestration (C++ PtzIncrementalOptimizer, all solves on the device)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import __graft_entry__ as ge
import host_util as hu

pkg = ge.load_package()
for n_views, opv in ((20, 100), (60, 300), (200, 500)):
    sc = pkg.synth.make_scene(1, n_views, opv)
    t0 = time.time(); tb = pkg.synth.make_match_table(sc); tg = time.time() - t0
    cam0 = np.zeros((tb.n_img, 15)); cam0[:, 0] = cam0[:, 1] = 1.0
    for rep in range(2):
        t0 = time.time()
        ok, cam, reg, events, nit = hu.incremental_solve(tb, cam0, max_iter=200)
        dt = time.time() - t0
    nba = sum(1 for e in events if e[0] == 2); nreg = sum(1 for e in events if e[0] == 1)
    ferr = np.abs(cam[reg, 0] / sc.cam_gt[reg, 0] - 1).max() if reg else float("nan")
    print(f"views {n_views} pairs {tb.n_pairs} matches {len(tb.q)} (table gen {tg:.1f}s): ok={ok} registered={len(reg)} "
          f"wall={dt*1e3:.1f} ms  BA calls={nba} (LM it {nit}) registrations={nreg}  views/s={len(reg)/dt:.1f} max f err={ferr:.2e}\n    ms: rank/BA(total,device)/reg(total,device) = {np.round(hu.incremental_solve.timing, 1)}", flush=True)
