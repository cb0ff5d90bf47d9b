import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import __graft_entry__ as ge
import host_util as hu
pkg = ge.load_package(); orc = ge.load_oracle()
import incremental_oracle as io
for seed, n, bi in ((1, 20, True), (3, 24, False), (6, 40, True), (8, 60, True)):
    sc = pkg.synth.make_scene(seed, n, 100)
    tb = pkg.synth.make_match_table(sc, bidirectional=bi)
    cam0 = np.zeros((tb.n_img, 15)); cam0[:, 0] = cam0[:, 1] = 1.0
    ok, cam, reg, events, nit = hu.incremental_solve(tb, cam0, max_iter=200)
    for mode in (orc.JAC_ANALYTIC, orc.JAC_NUMERIC):
        o = io.IncrementalOracle(tb, cam0, 200, jacobian_mode=mode); o.solve()
        oc = o.cam15()
        same = events == o.events
        rf = np.abs(cam[reg, 0] / oc[reg, 0] - 1).max() if reg == sorted(o.reg) else float("nan")
        rr = max(np.abs(orc.rodrigues(cam[i, 4:7]) - orc.rodrigues(oc[i, 4:7])).max() for i in reg) if reg == sorted(o.reg) else float("nan")
        print(seed, n, bi, "mode", mode, "events equal", same, "reg", len(reg), len(o.reg), "f rel", rf, "R abs", rr, flush=True)
        if rr > 1e-3:
            d = [np.abs(orc.rodrigues(cam[i, 4:7]) - orc.rodrigues(oc[i, 4:7])).max() for i in reg]
            print("   per-cam R diff >1e-3:", [(i, round(x, 3)) for i, x in zip(reg, d) if x > 1e-3][:10], "rvec gpu/oracle", cam[reg[np.argmax(d)], 4:7], oc[reg[np.argmax(d)], 4:7])
