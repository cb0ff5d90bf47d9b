import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as ge
from types import SimpleNamespace
pkg = ge.load_package(); orc = ge.load_oracle()
import incremental_oracle as io
sc = pkg.synth.make_scene(3, 24, 100)
tb = pkg.synth.make_match_table(sc, bidirectional=False)
cam0 = np.zeros((tb.n_img, 15)); cam0[:, 0] = cam0[:, 1] = 1.0
probs = []
class Cap(io.IncrementalOracle):
    def _bundle(self, ids):
        cand = sorted(ids); slot = {im: k for k, im in enumerate(cand)}
        uv, oc, orr, w = [], [], [], []
        for tid in sorted(self.tracks):
            tr = self.tracks[tid]
            views = [im for im in sorted(tr) if im in slot]
            if not views: continue
            for im in views:
                uv.append(self.kp[im][tr[im]]); oc.append(slot[im]); orr.append(len(w))
            w.append(float(len(tr)))
        camv = np.stack([io._cam_to_vec(self.cams[im]) for im in cand])
        ns = SimpleNamespace(obs_uv=np.asarray(uv, dtype=np.float32), obs_cam=np.asarray(oc, dtype=np.int32),
                             obs_ray=np.asarray(orr, dtype=np.int32), ray_weight=np.asarray(w), n_cam=len(cand), n_ray=len(w),
                             n_obs=len(oc), factor_type=0, cam_init=camv)
        ns.ray_init = self._pix2ray(cand, ns)
        probs.append(ns)
        return super()._bundle(ids)
o = Cap(tb, cam0, 200, jacobian_mode=orc.JAC_ANALYTIC); o.solve()
print("captured", len(probs), [p.n_cam for p in probs])
def run(k, iters=200):
    cam, ray, summ = pkg.api.ba_solve(probs[k], max_num_iterations=iters)
    return cam, ray, summ
base = {}
for k in range(len(probs)):
    base[k] = run(k)
# second pass in reverse order, third after a big unrelated solve
for order, name in ((list(reversed(range(len(probs)))), "reverse"), (list(range(len(probs))), "forward again")):
    for k in order:
        cam, ray, summ = run(k)
        if not (np.array_equal(cam, base[k][0]) and np.array_equal(ray, base[k][1]) and summ == base[k][2]):
            print(name, "problem", k, "n_cam", probs[k].n_cam, "DIFFERS: iters", summ["num_iterations"], base[k][2]["num_iterations"],
                  "max cam diff", np.abs(cam - base[k][0]).max())
            for it in (1, 2):
                c1, r1, s1 = run(k, it)
                print("    rerun with max_iter", it, "cost", s1["final_cost"])
print("done")
