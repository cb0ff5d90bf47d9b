#!/usr/bin/env python3
"""Generates tests/golden/lm_trajectories.json: LM trajectories of the reference-faithful CPU oracle
(central-difference Jacobians with Ceres' step rule, Ceres-1.14 trust-region policy) on seeded synthetic
problems -- cost per iteration, radius, accept/reject, termination, refined focal lengths.  These pin the
oracle against accidental drift and give the GPU tests a committed target besides the live oracle."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import __graft_entry__ as ge  # noqa: E402

pkg_synth = None


def main():
    import importlib.util
    spec = importlib.util.spec_from_file_location("synth_only", os.path.join(os.path.dirname(HERE), "ptz-calib_amd", "synth.py"))
    synth = importlib.util.module_from_spec(spec)
    sys.modules["synth_only"] = synth
    spec.loader.exec_module(synth)
    orc = ge.load_oracle()
    doc = {"_generator": "oracle/gen_golden_lm.py (oracle numeric-diff mode, 1 thread)", "ba": [], "krt": []}
    for name, args in [("c1_ptzray", dict(scene_id=0, n_views=20, obs_per_view=100)),
                       ("c1_ptzraydist", dict(scene_id=3, n_views=20, obs_per_view=100, factor_type=1)),
                       ("c3like_seed5", dict(scene_id=5, n_views=60, obs_per_view=300))]:
        sc = synth.make_scene(**args)
        cam, ray, _, s, tr = orc.ba_solve(sc, jacobian_mode=orc.JAC_NUMERIC, trace=True, num_threads=1)
        doc["ba"].append(dict(name=name, scene=args, n_obs=sc.n_obs, n_ray=sc.n_ray, summary=s, cost=tr.cost.tolist(),
                              radius=tr.radius.tolist(), accepted=tr.accepted.tolist(), focal=cam[:, 0].tolist(),
                              k1=cam[:, 10].tolist()))
        print(name, s["termination_type"], s["num_iterations"], s["final_cost"])
    for ftype in (0, 1):
        rb = synth.make_reloc_batch(8, 128, seed_id=ftype, factor_type=ftype)
        for q in range(rb.n_query):
            sl = slice(rb.match_ptr[q], rb.match_ptr[q + 1])
            loc0 = orc.krt_world_to_local(rb.cam_ref[q], rb.cam_init[q])
            loc, s, tr = orc.krt_solve(rb.uv_ref[sl], rb.uv_cur[sl], rb.cam_ref[q], loc0, factor_type=ftype,
                                       jacobian_mode=orc.JAC_NUMERIC, trace=True)
            doc["krt"].append(dict(factor_type=ftype, seed_id=ftype, query=q, summary=s, cost=tr.cost.tolist(),
                                   accepted=tr.accepted.tolist(), cam_local=loc.tolist(),
                                   accepted_by_gates=bool(orc.krt_check(s, loc, 100.0))))
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "lm_trajectories.json")
    with open(out, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote", out)


if __name__ == "__main__":
    main()
