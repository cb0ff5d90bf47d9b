#!/usr/bin/env python3
"""Generates tests/golden/lm_trajectories_variants.json: LM trajectories of the reference-faithful CPU oracle (central-difference
Jacobians, Ceres-1.14 trust-region policy) for the model variants added after the first fixture -- PTZRayFxfyDist, the
georeferencing solve (2D-3D annotations + T_l_w), shared intrinsics, and the single-view LM with 2D-3D constraints.
Same purpose as gen_golden_lm.py: pins the oracle against drift and gives the GPU tests a committed target."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import __graft_entry__ as ge  # noqa: E402


def main():
    synth = ge.load_package().synth
    orc = ge.load_oracle()
    doc = {"_generator": "oracle/gen_golden_lm_variants.py (oracle numeric-diff mode, 1 thread)", "ba": [], "krt_2d3d": []}
    cases = [("fxfydist", dict(scene_id=3, n_views=20, obs_per_view=100, factor_type=2), False, 1.015),
             ("georef_ptzray", dict(scene_id=2, n_views=20, obs_per_view=100, factor_type=0), True, None),
             ("georef_fxfydist", dict(scene_id=4, n_views=20, obs_per_view=100, factor_type=2), True, 1.015),
             ("shared_intrinsics_dist", dict(scene_id=6, n_views=24, obs_per_view=100, factor_type=1, n_intrinsics_groups=3), False, None)]
    for name, args, annotated, fy_scale in cases:
        sc = synth.make_scene(**args)
        if annotated:
            sc = synth.add_annotations(sc)
        if fy_scale is not None:
            sc.cam_init = sc.cam_init.copy()
            sc.cam_init[:, 1] = sc.cam_init[:, 0] * fy_scale
        kw = dict(obs3d=sc.obs3d, tlw0=sc.tlw_init) if annotated else {}
        cam, ray, tlw, s, tr = orc.ba_solve(sc, jacobian_mode=orc.JAC_NUMERIC, trace=True, num_threads=1, **kw)
        doc["ba"].append(dict(name=name, scene=args, annotated=annotated, fy_scale=fy_scale, n_obs=sc.n_obs, n_ray=sc.n_ray, summary=s,
                              cost=tr.cost.tolist(), accepted=tr.accepted.tolist(), focal=cam[:, 0].tolist(), fy=cam[:, 1].tolist(),
                              k1=cam[:, 10].tolist(), tlw=tlw.tolist()))
        print(name, s["termination_type"], s["num_iterations"], s["final_cost"])
    for ftype in (0, 3):
        rb = synth.add_reloc_points(synth.make_reloc_batch(6, 96, seed_id=40 + ftype, factor_type=ftype), n_pt=10)
        for q in range(rb.n_query):
            sl = slice(rb.match_ptr[q], rb.match_ptr[q + 1]); ps = slice(rb.point_ptr[q], rb.point_ptr[q + 1])
            loc0 = orc.krt_world_to_local(rb.cam_ref[q], rb.cam_init[q])
            Xl = orc.krt_point_to_local(rb.cam_ref[q], rb.pts3d[ps])
            loc, s, tr = orc.krt_solve(rb.uv_ref[sl], rb.uv_cur[sl], rb.cam_ref[q], loc0, factor_type=ftype, pts2d=rb.pts2d[ps],
                                       pts3d_local=Xl, jacobian_mode=orc.JAC_NUMERIC, trace=True)
            doc["krt_2d3d"].append(dict(factor_type=ftype, seed_id=40 + ftype, query=q, summary=s, cost=tr.cost.tolist(),
                                        accepted=tr.accepted.tolist(), cam_local=loc.tolist(),
                                        accepted_by_gates=bool(orc.krt_check(s, loc, 100.0))))
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "lm_trajectories_variants.json")
    with open(out, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote", out)


if __name__ == "__main__":
    main()
