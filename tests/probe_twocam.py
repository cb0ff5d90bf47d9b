import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from types import SimpleNamespace
pkg = ge.load_package(); orc = ge.load_oracle()
import incremental_oracle as io
sc = pkg.synth.make_scene(1, 20, 100)
mt = pkg.synth.make_match_table(sc)
cam0 = np.zeros((20, 15)); cam0[:, 0] = cam0[:, 1] = 1

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
        ns.ray_init = orc.pix2ray(ns, camv)
        self.captured = ns
        raise StopIteration

o = Cap(mt, cam0, 200)
try:
    o.solve()
except StopIteration:
    pass
ns = o.captured
print("n_obs", ns.n_obs, "n_ray", ns.n_ray, "cams", ns.cam_init[:, :7])
for mode in (orc.JAC_ANALYTIC, orc.JAC_NUMERIC):
    cam, ray, _, summ, tr = orc.ba_solve(ns, jacobian_mode=mode, trace=True)
    print("oracle", mode, summ)
    print(" cost", tr.cost, "\n radius", getattr(tr, "radius", None), "\n ok", getattr(tr, "step_ok", None))
cam, ray, summ = pkg.api.ba_solve(ns)
print("gpu", summ)
