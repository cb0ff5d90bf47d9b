"""CPU tests of the C++ host mirror (ptz-calib_amd/host): track builder against the reference-generated goldens,
problem packing (ordering, weights, candidate subsets, Pix2Ray) against the oracle."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest

import host_util as hu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(GOLD, "tracks_reference.json")))["cases"]


@pytest.mark.parametrize("name", sorted(CASES.keys()))
def test_host_tracks_match_reference_golden(name):
    case = CASES[name]
    pairs = [(i, j, [tuple(m) for m in ms]) for i, j, ms in case["pairs"]]
    got = hu.tracks_build(pairs, case["min_track_length"])
    want = {int(t): {int(i): f for i, f in v.items()} for t, v in case["tracks"].items()}
    assert got == want  # bit-exact track ids and membership


def test_host_tracks_equal_oracle_on_random_graphs(orc):
    rng = np.random.default_rng(21)
    for _ in range(20):
        n_img = int(rng.integers(4, 16))
        pairs = []
        for _ in range(int(rng.integers(1, 50))):
            i, j = rng.choice(n_img, 2, replace=False)
            ms = [(int(t), int(t if rng.random() < 0.9 else rng.integers(0, 40))) for t in rng.integers(0, 40, int(rng.integers(0, 30)))]
            pairs.append((int(i), int(j), ms))
        ml = int(rng.integers(2, 5))
        assert hu.tracks_build(pairs, ml) == orc.tracks_build(pairs, ml)


def test_packing_order_weights_and_pix2ray(pkg, orc, scene_c1):
    sc = scene_c1
    kps, plist = hu.scene_to_features_matches(sc)
    ok, cam, err, summ, pk = hu.ptzray_solve(kps, plist, sc.cam_init, on_device=False)
    tracks = orc.tracks_build(plist, 4)
    assert len(pk["ray_weight"]) == len(tracks) == sc.n_ray
    # residual order = (track id asc, image id asc) (AddConstraints2d2d, ptzray_optimizer.cc:801-850)
    assert np.all(np.diff(pk["obs_ray"]) >= 0)
    a = 0
    for r, (tid, tr) in enumerate(sorted(tracks.items())):
        imgs = sorted(tr.keys())
        assert pk["ray_weight"][r] == len(imgs)
        for img in imgs:
            assert pk["obs_ray"][a] == r and pk["obs_cam"][a] == img
            assert np.array_equal(pk["obs_uv"][a], np.asarray(kps[img][tr[img]], dtype=np.float32))
            a += 1
    assert a == sc.n_obs
    assert np.allclose(pk["cam"], sc.cam_init, rtol=0, atol=1e-12)  # ToVector(FromVector(v)) round trip
    ns = SimpleNamespace(obs_uv=pk["obs_uv"], obs_cam=pk["obs_cam"], obs_ray=pk["obs_ray"], ray_weight=pk["ray_weight"],
                         n_cam=sc.n_cam, n_ray=len(pk["ray_weight"]), factor_type=0)
    assert np.abs(pk["ray"] - orc.pix2ray(ns, pk["cam"])).max() < 1e-13


def test_packing_candidate_subset_keeps_full_track_weight(pkg, orc, scene_c1):
    """cam_ids subset: only candidate views produce residuals, but the ScaledLoss weight stays the FULL track
    length (ptzray_optimizer.cc:805) and Pix2Ray averages only candidate views (:780-781)."""
    sc = scene_c1
    kps, plist = hu.scene_to_features_matches(sc)
    cand = [2, 3, 5, 6, 7, 9]
    ok, cam, err, summ, pk = hu.ptzray_solve(kps, plist, sc.cam_init, cand_ids=cand, on_device=False)
    tracks = orc.tracks_build(plist, 4)
    assert list(pk["cam_image"]) == sorted(cand)
    kept = [(tid, tr) for tid, tr in sorted(tracks.items()) if any(i in cand for i in tr)]
    assert len(pk["ray_weight"]) == len(kept)
    for r, (tid, tr) in enumerate(kept):
        assert pk["ray_weight"][r] == len(tr)  # full length, not the candidate count
    assert set(np.unique(pk["obs_cam"])) <= set(range(len(cand)))
    n_expected = sum(sum(1 for i in tr if i in cand) for _, tr in kept)
    assert len(pk["obs_cam"]) == n_expected
