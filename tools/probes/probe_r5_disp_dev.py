import sys, os, numpy as np, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
pkg = ge.load_package()
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oracle"))
import oracle_py as orc
for seed, views, obs in ((2, 20, 100), (5, 40, 200)):
  sc = pkg.synth.make_scene(seed, views, obs, factor_type=3)
  for ft in (1e-6, 1e-10, 1e-14):
    kw = dict(function_tolerance=ft, max_num_iterations=1000)
    cam, ray, summ, _, disp = pkg.api.ba_solve_disp(sc, **kw)
    od = np.zeros(3)
    t = time.time()
    ocam, oray, _, osumm, _ = orc.ba_solve(sc, jacobian_mode=orc.JAC_NUMERIC, disp=od, num_threads=8, **kw)
    f = ocam[:, 0]
    dl = disp[0] + disp[1] * f + disp[2] * f * f; odl = od[0] + od[1] * f + od[2] * f * f
    print(seed, ft, "iters dev/num", summ["num_iterations"], osumm["num_iterations"], "term", summ["termination_type"], osumm["termination_type"],
          "cost rel %.3e" % (abs(summ["final_cost"] - osumm["final_cost"]) / osumm["final_cost"]),
          "df/f %.3e" % np.abs(cam[:, 0] / ocam[:, 0] - 1).max(), "dk1 %.3e" % np.abs(cam[:, 10] - ocam[:, 10]).max(),
          "ddelta %.3e (delta ~ %.3e)" % (np.abs(dl - odl).max(), np.abs(odl).max()), "%.1fs" % (time.time() - t), flush=True)
