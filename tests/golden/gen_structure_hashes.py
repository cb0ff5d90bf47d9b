#!/usr/bin/env python3
"""Fixture: FNV-1a hashes of the host-side structure of ptz_ba_batch_create (internal ray order, observation arrays, camera-major
lists, camera-pair entry lists, k_schur's runs) for a few synthetic scenes, as ptz_debug_host_structure reports them.  Written
with the library as of round 3 after its output had been compared, hash for hash, with the round-2 builder on the same scenes
(including one whose tracks are not camera-ascending).  The order of the entries fixes the order of every sum in k_schur, so a
later, faster builder has to reproduce these or the bits of every solve move.  Runs without a GPU."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

CASES = [dict(scene_id=40, n_views=20, obs_per_view=100), dict(scene_id=41, n_views=60, obs_per_view=300, factor_type=1),
         dict(scene_id=42, n_views=170, obs_per_view=500), dict(scene_id=43, n_views=200, obs_per_view=500),
         dict(scene_id=44, n_views=33, obs_per_view=77, factor_type=2), dict(scene_id=45, n_views=8, obs_per_view=40)]


def shuffled_tracks(pkg, seed=77):
    """a scene whose tracks are NOT camera-ascending (the general path of the entry builder)"""
    import copy
    sc = pkg.synth.make_scene(seed, 40, 200)
    rng = np.random.default_rng(5)
    sh = copy.copy(sc)
    oc, ou = sc.obs_cam.copy(), sc.obs_uv.copy()
    ptr = np.flatnonzero(np.r_[1, np.diff(sc.obs_ray), 1])
    for a, b in zip(ptr[:-1], ptr[1:]):
        perm = rng.permutation(b - a)
        oc[a:b] = oc[a:b][perm]; ou[a:b] = ou[a:b][perm]
    sh.obs_cam, sh.obs_uv = oc, ou
    return sh


def structure_hash(pkg, scenes, threads):
    keep = []
    probs = (pkg.api.BaProblem * len(scenes))(*[pkg.api._pack_problem(s, keep) for s in scenes])
    ms, h = C.c_double(), C.c_uint64()
    rc = pkg.api.lib().ptz_debug_host_structure(len(scenes), probs, threads, 1, C.byref(ms), C.byref(h))
    assert rc == 0, rc
    return int(h.value)


def all_hashes(pkg):
    scenes = [pkg.synth.make_scene(**c) for c in CASES] + [shuffled_tracks(pkg)]
    out = {"single": [structure_hash(pkg, [s], 1) for s in scenes]}
    plain = [s for s in scenes if s.factor_type == 0]
    out["batch_of_plain"] = structure_hash(pkg, plain, 3)
    return out


if __name__ == "__main__":
    pkg = ge.load_package()
    json.dump(all_hashes(pkg), open(os.path.join(ROOT, "tests", "golden", "structure_hashes.json"), "w"), indent=1)
