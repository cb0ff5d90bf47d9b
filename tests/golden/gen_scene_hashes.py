#!/usr/bin/env python3
"""Fixture: SHA-256 of the packed arrays of synthetic scenes (synth.make_scene) for a spread of seeds and shapes, written by
the generator as it stood at the end of round 2 (dense N x P visibility tables).  tests/test_cpu_abi_host.py holds every later
generator to them: the scenes behind the committed minima / trajectory fixtures and behind bench.py must not move by a bit."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

CASES = [
    dict(scene_id=0, n_views=20, obs_per_view=100), dict(scene_id=1, n_views=20, obs_per_view=100, factor_type=1),
    dict(scene_id=2, n_views=20, obs_per_view=100, factor_type=3), dict(scene_id=5, n_views=60, obs_per_view=300),
    dict(scene_id=6, n_views=60, obs_per_view=300), dict(scene_id=7, n_views=60, obs_per_view=300),
    dict(scene_id=11, n_views=60, obs_per_view=300, width=1280, height=720, pan_range_deg=120.0),
    dict(scene_id=12, n_views=60, obs_per_view=300, factor_type=1, width=1280, height=720, pan_range_deg=120.0),
    dict(scene_id=100, n_views=60, obs_per_view=300, factor_type=1, width=1280, height=720, pan_range_deg=120.0),
    dict(scene_id=0, n_views=200, obs_per_view=500), dict(scene_id=3, n_views=200, obs_per_view=500),
    dict(scene_id=999, n_views=200, obs_per_view=500), dict(scene_id=0, n_views=200, obs_per_view=20),
    dict(scene_id=9, n_views=30, obs_per_view=120), dict(scene_id=4, n_views=36, obs_per_view=100),
    dict(scene_id=21, n_views=44, obs_per_view=120), dict(scene_id=3, n_views=24, obs_per_view=100, n_intrinsics_groups=3),
    dict(scene_id=8, n_views=10, obs_per_view=60), dict(scene_id=2, n_views=330, obs_per_view=40, pan_range_deg=340.0),
]


def scene_hash(sc):
    h = hashlib.sha256()
    for a in (sc.obs_uv, sc.obs_cam, sc.obs_ray, sc.ray_weight, sc.cam_gt, sc.cam_init, sc.ray_gt, sc.ray_init):
        h.update(a.tobytes())
    h.update(repr((sc.n_cam, sc.n_ray, sc.n_obs)).encode())
    return h.hexdigest()


if __name__ == "__main__":
    pkg = ge.load_package()
    out = [dict(args=c, n_ray=int(sc.n_ray), n_obs=int(sc.n_obs), sha256=scene_hash(sc))
           for c in CASES for sc in [pkg.synth.make_scene(**c)]]
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scene_hashes.json"), "w"), indent=1)
    print(len(out), "scenes hashed")
