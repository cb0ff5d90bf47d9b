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


def test_host_tracks_sparse_feature_ids_take_the_sorted_path(orc):
    """Feature ids far apart (beyond the dense rank table of TracksBuilder::Build) go through sort + unique + binary search;
    both paths number the nodes identically, so the track ids agree with the oracle either way."""
    rng = np.random.default_rng(5)
    feats = [int(x) for x in rng.choice(1 << 30, 40, replace=False)]
    pairs = []
    for i in range(6):
        for j in range(i + 1, 6):
            ms = [(feats[k], feats[k]) for k in rng.choice(40, 12, replace=False)]
            pairs.append((i, j, ms))
    assert hu.tracks_build(pairs, 3) == orc.tracks_build(pairs, 3)
    small = [(i, j, [(a % 50, b % 50) for a, b in ms]) for i, j, ms in pairs]
    assert hu.tracks_build(small, 3) == orc.tracks_build(small, 3)


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


# ------------------------------------------------------------------------------------------------ EPnP initialisation
def _rodrigues(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def _pnp_case(rng, n, planar, noise, k1=0.0):
    R = _rodrigues(rng.normal(size=3) * 0.8)
    Xc = np.c_[rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(4, 12, n)]
    if planar:
        nrm = _rodrigues(rng.normal(size=3) * 0.5) @ np.array([0, 0, 1.0])
        a = np.cross(nrm, [1, 0, 0]); a /= np.linalg.norm(a); b = np.cross(nrm, a)
        ab = rng.uniform(-3, 3, (n, 2))
        Xc = np.array([0, 0, 9.0]) + ab[:, :1] * a + ab[:, 1:] * b
    t = rng.normal(size=3) * 3
    Xw = (Xc - t) @ R
    x, y = Xc[:, 0] / Xc[:, 2], Xc[:, 1] / Xc[:, 2]
    rad = 1 + k1 * (x * x + y * y)
    uv = np.c_[2000 * x * rad + 960, 2000 * y * rad + 540] + rng.normal(size=(n, 2)) * noise
    return R, t, Xw, uv


@pytest.mark.parametrize("planar", [False, True])
def test_epnp_recovers_exact_pose(planar):
    """SolvePnPEPnP (replacement for cv::solvePnP(..., SOLVEPNP_EPNP), ptzray_optimizer.cc:572): noise-free correspondences
    give the pose back to round-off, for general and for coplanar (pitch) points, with and without k1."""
    import host_util as hu
    rng = np.random.default_rng(11)
    K = np.array([[2000, 0, 960], [0, 2000, 540], [0, 0, 1.0]])
    for trial in range(40):
        n = int(rng.integers(4 if not planar else 5, 30))
        k1 = 0.0 if trial % 2 == 0 else 0.05
        R, t, Xw, uv = _pnp_case(rng, n, planar, 0.0, k1)
        ok, Re, te = hu.epnp(Xw, uv.astype(np.float64), K, np.array([k1, 0, 0, 0, 0.0]))
        assert ok
        # float32 pixels are the only error source
        assert np.abs(Re - R).max() < 2e-4 and np.abs(te - t).max() < 5e-3
        assert abs(np.linalg.det(Re) - 1) < 1e-12


def test_epnp_with_noise_and_degenerate_inputs():
    import host_util as hu
    rng = np.random.default_rng(12)
    K = np.array([[2000, 0, 960], [0, 2000, 540], [0, 0, 1.0]])
    for planar in (False, True):
        worst = 0.0
        for _ in range(40):
            R, t, Xw, uv = _pnp_case(rng, 12, planar, 0.5)
            ok, Re, te = hu.epnp(Xw, uv, K, np.zeros(5))
            assert ok
            worst = max(worst, np.degrees(np.arccos(np.clip((np.trace(Re @ R.T) - 1) / 2, -1, 1))))
        assert worst < (1.0 if planar else 0.3)
    R, t, Xw, uv = _pnp_case(rng, 3, False, 0.0)
    assert not hu.epnp(Xw, uv, K, np.zeros(5))[0]                       # fewer than 4 points
    line = np.outer(np.linspace(0, 1, 6), [1.0, 2.0, 0.5]) + [0, 0, 8]
    assert not hu.epnp(line, np.zeros((6, 2)), K, np.zeros(5))[0]       # collinear points


@pytest.mark.parametrize("ftype", [0, 1])
def test_georef_packing_and_tlw_initialisation(pkg, ftype):
    """PTZRayOptimizer with annotations, stages before the device solve: SetInitTransLocalToWorld (ptzray_optimizer.cc:562-633)
    from the first annotated candidate view, 2D-3D blocks in camera order (:894-917)."""
    import host_util as hu
    sc = pkg.synth.add_annotations(pkg.synth.make_scene(2, 20, 100, factor_type=ftype))
    kps, plist = hu.scene_to_features_matches(sc)
    ann = (sc.obs3d["cam"], sc.obs3d["uv"], sc.obs3d["xyz"])
    # georeferencing runs on cameras already refined by the bundle adjustment (run_ptz_ba.cc:131-155): use the true ones here
    ok, cam, err, summ, pk = hu.ptzray_solve(kps, plist, sc.cam_gt, ftype=ftype, annotations=ann, on_device=False)
    assert not ok and pk["tlw_ok"]
    Rg, Ri = _rodrigues(sc.tlw_gt[:3]), _rodrigues(pk["tlw_init"][:3])
    assert np.degrees(np.arccos(np.clip((np.trace(Ri @ Rg.T) - 1) / 2, -1, 1))) < 1.0
    assert np.abs(-Ri.T @ pk["tlw_init"][3:] - np.array([3.0, -45.0, 15.0])).max() < 2.0
    n3 = len(sc.obs3d["cam"])
    assert pk["n_obs3d"] == n3
    # annotations that project behind the camera fail the gates (:581-586): T_l_w stays zero, the solve still proceeds
    # mismatched annotations (pixels shuffled against the points): the 300 px gate (:602-605) rejects every view
    far = (sc.obs3d["cam"], sc.obs3d["uv"][::-1].copy(), sc.obs3d["xyz"])
    ok2, _, _, _, pk2 = hu.ptzray_solve(kps, plist, sc.cam_gt, ftype=ftype, annotations=far, on_device=False)
    assert not ok2 and not pk2["tlw_ok"] and np.array_equal(pk2["tlw_init"], np.zeros(6))
    # candidate subset without any annotated view: no PnP, T_l_w = 0
    cands = [i for i in range(sc.n_cam) if i not in set(sc.obs3d["cam"].tolist())]
    ok3, _, _, _, pk3 = hu.ptzray_solve(kps, plist, sc.cam_gt, cand_ids=cands, ftype=ftype, annotations=ann, on_device=False)
    assert not ok3 and not pk3["tlw_ok"] and np.array_equal(pk3["tlw_init"], np.zeros(6))


# ------------------------------------------------------------------------------------------- PTZ-IBA orchestration (CPU part)
def _orchestration_oracle():
    import __graft_entry__ as ge
    ge.load_oracle()
    import incremental_oracle
    return incremental_oracle


@pytest.mark.parametrize("bidirectional", [False, True])
def test_incremental_seed_ranking_matches_restatement(pkg, bidirectional):
    """Without a device every bundle adjustment fails, so PtzIncrementalOptimizer::Solve walks through 50 seed pairs and
    returns false (ptz_incremental_optimizer.cc:44-63).  The SEQUENCE of seed pairs exercises FindFirstInitialImage,
    FindSecondInitialImage, CalPixelDiff and the try-once bookkeeping (:148-244, :298-320), including the std::sort order of
    tied scores, and must equal the restatement's."""
    io = _orchestration_oracle()
    sc = pkg.synth.make_scene(3, 24, 100)
    tb = pkg.synth.make_match_table(sc, bidirectional=bidirectional)
    cam0 = np.zeros((tb.n_img, 15)); cam0[:, 0] = cam0[:, 1] = 1.0
    ok, cam, reg, events, nit = hu.incremental_solve(tb, cam0, max_iter=50)
    assert not ok and nit == 0

    class NoSolver(io.IncrementalOracle):
        def _bundle(self, ids):
            self.events.append((2, len(ids), 0, 0))
            return False

    o = NoSolver(tb, cam0, 50)
    assert not o.solve()
    assert [e for e in events if e[0] == 0] == [e for e in o.events if e[0] == 0]
    assert len([e for e in events if e[0] == 0]) == 50 and events == o.events
    # manual seeds (SetSeedImageId, :133-137) replace the first-image ranking
    ok2, _, _, ev2, _ = hu.incremental_solve(tb, cam0, max_iter=50, seeds=[5])
    o2 = NoSolver(tb, cam0, 50); o2.seeds = [5]
    assert not o2.solve() and not ok2 and ev2 == o2.events and all(e[1] == 5 for e in ev2 if e[0] == 0)


def test_incremental_invalid_inputs(pkg):
    sc = pkg.synth.make_scene(3, 8, 60)
    tb = pkg.synth.make_match_table(sc)
    cam0 = np.zeros((tb.n_img, 15)); cam0[:, 0] = cam0[:, 1] = 1.0
    assert not hu.incremental_solve(tb, cam0, max_iter=0)[0]  # CheckValid (:140-146)


def test_rodrigues_roundtrip_near_pi_and_scaled_matrices(orc):
    """cv::Rodrigues (matrix -> vector) as restated in the oracle and in the host Camera class: the theta ~ pi branch picks
    the axis signs from R01, R02 (and R12 when x is the smallest component); a scaled, noisy near-rotation (K_j^-1 H K_i of
    the registration step) is replaced by its polar factor first."""
    rng = np.random.default_rng(5)
    for th in [0.1, 1.0, 3.0, 3.1415, np.pi - 3e-6, np.pi - 1e-7, np.pi - 1e-9, np.pi, 3.3, 4.0]:
        for _ in range(25):
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
            v = np.zeros(15); v[0] = v[1] = 1000.0; v[4:7] = ax * th
            out = np.zeros(15)
            hu.lib().ptzh_camera_roundtrip(hu._p(v), hu._p(out), None)
            Ri = orc.rodrigues(v[4:7])
            # the special branch is entered within 1e-5 of pi, where the axis sign is only known up to that angle (as in OpenCV)
            tol = 3e-5 if abs(th - np.pi) < 1.1e-5 else 1e-9
            assert np.abs(orc.rodrigues(out[4:7]) - Ri).max() < tol
            assert np.abs(orc.rodrigues(orc.rodrigues_inv(Ri)) - Ri).max() < tol
            assert np.abs(pkg_synth().rodrigues(pkg_synth().rodrigues_inv(Ri)) - Ri).max() < tol
            M = Ri * rng.uniform(0.3, 3.0) + rng.normal(size=(3, 3)) * 1e-3
            U, _, Vt = np.linalg.svd(M)
            r1 = np.zeros(3)
            hu.lib().ptzh_rodrigues_inv(hu._p(np.ascontiguousarray(M.reshape(9))), hu._p(r1))
            P = U @ Vt
            if abs(np.arccos(np.clip((np.trace(P) - 1) / 2, -1, 1)) - np.pi) < 2e-5:
                continue
            assert np.abs(orc.rodrigues(r1) - P).max() < 1e-9
            assert np.abs(orc.rodrigues(orc.rodrigues_inv(M)) - P).max() < 1e-9


def pkg_synth():
    import __graft_entry__ as ge
    return ge.load_package().synth
