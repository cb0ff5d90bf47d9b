"""PTZRayDistDisp on the device (-m gpu): factor type 3 of PTZRayOptimizer (ptzray_optimizer.cc:195-265; with annotations
Reproj2d3dDispFactor :334-396; the one displacement block disp_param_ :655 shared by every residual).

The reference differentiates this functor with central differences whose step for d2 = 0 (sqrt(eps)) is far outside the
linear range of d2 * fx^2 (tests/test_cpu_oracle.py::test_displacement_variant_restatement), so the parity target is the
oracle in closed-form mode: same functor bits, same Jacobians to round-off, same LM bookkeeping.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


def _relative_rotations(orc, cam):
    R = [orc.rodrigues(c[4:7]) for c in cam]
    return np.stack([r @ R[0].T for r in R])


def test_linearize_dist_disp_vs_oracle(pkg, orc):
    """One linearisation at a non-zero displacement: cost, U, g_c, V, g_r and the W = Jc^T Jr rows against the oracle's
    closed-form linearisation.  Device columns [f, k1, r1, r2, r3, d0, d1, d2]; the oracle carries the reference's fy column
    (always zero for the 2D-2D functor) as well."""
    sc = pkg.synth.make_scene(2, 20, 100, factor_type=3)
    cam = sc.cam_init.copy(); cam[:, 10] = 0.02
    ray = sc.ray_init * 1.2
    d = np.array([0.05, 2e-5, -1e-9])
    b = pkg.api.BaBatch([sc]); b.set_state([cam], [ray]); b.set_disp([d])
    assert b.nc == 8 and b.nw == 8
    g = b.linearize(0)
    b.close()
    o = orc.ba_linearize(sc, cam, ray, jacobian_mode=orc.JAC_ANALYTIC, disp=d)
    sel = [0, 2, 3, 4, 5, 6, 7, 8]
    assert abs(g["cost"] - o["cost"]) / o["cost"] < 1e-12
    assert _rel(g["V"], o["V"]) < 1e-11 and _rel(g["g_r"], o["g_r"]) < 1e-11
    for k in range(8):  # per column: the displacement columns differ by orders of magnitude
        assert _rel(g["W"][:, k], o["W"][:, sel[k]]) < 1e-10, k
        if k < 5:
            assert _rel(g["g_c"][:, k], o["g_c"][:, sel[k]]) < 1e-9, k
        else:  # the oracle keeps the gradient of the shared block at camera 0's slots; the device folds its per-camera copies later
            assert not o["g_c"][1:, sel[k]].any()
            assert abs(g["g_c"][:, k].sum() - o["g_c"][0, sel[k]]) < 1e-9 * np.abs(g["g_c"][:, k]).sum(), k
        for l in range(8):
            assert _rel(g["U"][:, k, l], o["U"][:, sel[k], sel[l]]) < 1e-10, (k, l)


@pytest.mark.parametrize("seed,views,obs", [(2, 20, 100), (5, 40, 200)])
def test_ba_dist_disp_parity(pkg, orc, seed, views, obs):
    """The solve against the oracle in closed-form mode: termination, LM bookkeeping, cost, focal lengths, k1, gauge-free
    rotations and the displacement's effect delta(f) = d0 + d1 f + d2 f^2 (the three coefficients themselves are only
    determined through it on a rig whose focal lengths cluster)."""
    sc = pkg.synth.make_scene(seed, views, obs, factor_type=3)
    cam, ray, summ, _, disp = pkg.api.ba_solve_disp(sc)
    od = np.zeros(3)
    ocam, oray, _, osumm, _ = orc.ba_solve(sc, jacobian_mode=orc.JAC_ANALYTIC, disp=od, num_threads=4)
    assert summ["termination_type"] == osumm["termination_type"] == 0
    assert summ["num_iterations"] == osumm["num_iterations"]
    assert summ["num_successful_steps"] == osumm["num_successful_steps"]
    assert abs(summ["initial_cost"] - osumm["initial_cost"]) / osumm["initial_cost"] < 1e-12
    assert abs(summ["final_cost"] - osumm["final_cost"]) / osumm["final_cost"] < 1e-8
    assert _rel(cam[:, 0], ocam[:, 0]) < 1e-6
    assert np.abs(cam[:, 10] - ocam[:, 10]).max() < 1e-6
    assert np.abs(_relative_rotations(orc, cam) - _relative_rotations(orc, ocam)).max() < 1e-6
    f = ocam[:, 0]
    delta = disp[0] + disp[1] * f + disp[2] * f * f
    odelta = od[0] + od[1] * f + od[2] * f * f
    assert disp.any() and np.abs(delta - odelta).max() < 1e-6 * max(1.0, np.abs(odelta).max())
    # nothing but fx, k1, rvec moves in the camera vector (fy is not a parameter of the 2D-2D functor's problem here)
    assert np.array_equal(cam[:, [1, 2, 3, 7, 8, 9, 11, 12, 13, 14]], sc.cam_init[:, [1, 2, 3, 7, 8, 9, 11, 12, 13, 14]])


@pytest.mark.parametrize("seed,views,obs", [(2, 20, 100), (5, 40, 200)])
def test_ba_dist_disp_deviation_from_the_numeric_reference_has_a_number(pkg, orc, seed, views, obs):
    """What the documented deviation of this variant amounts to.  The reference differentiates PTZRayDistDisp numerically, and its
    central difference for d2 is a percent off (module docstring): the device (closed form) and the reference-faithful oracle
    (JAC_NUMERIC) therefore walk different paths -- and this model has a flat valley in (f, k1, delta(f)): focal length and the z
    displacement trade against each other, so different paths stop at different points of it.  Measured (probe_r5_disp_dev.py), default
    tolerances: seed 5 agrees to 5e-6 in f / 2e-6 in k1 / 5e-6 in delta after 19 iterations on both sides; seed 2 stops after 114
    (device) against 7 (numeric) iterations 8 % apart in f, 0.03 in k1, 0.11 in delta -- with final costs 1.6e-4 apart, the device's
    the LOWER one.  Tightening function_tolerance does not close the gap (both run along the valley until max_num_iterations).
    The bound this test can honestly hold: same termination type, the device's minimum is at least as good as the reference's to
    5e-4 in cost, and the parameters stay inside the valley's measured extent."""
    sc = pkg.synth.make_scene(seed, views, obs, factor_type=3)
    cam, ray, summ, _, disp = pkg.api.ba_solve_disp(sc)
    od = np.zeros(3)
    ocam, oray, _, osumm, _ = orc.ba_solve(sc, jacobian_mode=orc.JAC_NUMERIC, disp=od, num_threads=8)
    assert summ["termination_type"] == osumm["termination_type"] == 0
    assert summ["final_cost"] <= osumm["final_cost"] * (1 + 5e-4)
    assert abs(summ["final_cost"] - osumm["final_cost"]) / osumm["final_cost"] < 5e-4
    f = ocam[:, 0]
    delta = disp[0] + disp[1] * f + disp[2] * f * f
    odelta = od[0] + od[1] * f + od[2] * f * f
    df, dk1, dd = np.abs(cam[:, 0] / f - 1).max(), np.abs(cam[:, 10] - ocam[:, 10]).max(), np.abs(delta - odelta).max()
    assert df < 0.15 and dk1 < 0.06 and dd < 0.25, (df, dk1, dd)
    if seed == 5:  # where both sides take the same number of steps the variant meets the north-star tolerance scale
        assert summ["num_iterations"] == osumm["num_iterations"] and df < 2e-5 and dk1 < 1e-5 and dd < 2e-5, (df, dk1, dd)


def test_ba_dist_disp_with_annotations(pkg, orc):
    """With georeferencing residuals (Reproj2d3dDispFactor): fy live, the T_l_w block in the reduced system, the displacement
    block shared by the 2D-2D and the 2D-3D residuals."""
    sc = pkg.synth.make_scene(4, 24, 120, factor_type=3)
    pkg.synth.add_annotations(sc)
    cam, ray, summ, tlw, disp = pkg.api.ba_solve_disp(sc)
    od = np.zeros(3)
    ocam, oray, otlw, osumm, _ = orc.ba_solve(sc, obs3d=sc.obs3d, tlw0=sc.tlw_init, jacobian_mode=orc.JAC_ANALYTIC, disp=od, num_threads=4)
    assert summ["termination_type"] == osumm["termination_type"]
    assert summ["num_iterations"] == osumm["num_iterations"]
    assert abs(summ["final_cost"] - osumm["final_cost"]) / osumm["final_cost"] < 1e-8
    assert _rel(cam[:, 0], ocam[:, 0]) < 1e-6
    ann = np.unique(sc.obs3d["cam"])
    assert _rel(cam[ann, 1], ocam[ann, 1]) < 1e-6 and np.all(cam[ann, 1] != sc.cam_init[ann, 1])
    Rlw, oRlw = orc.rodrigues(tlw[:3]), orc.rodrigues(otlw[:3])
    for i in range(sc.n_cam):
        assert np.abs(orc.rodrigues(cam[i, 4:7]) @ Rlw - orc.rodrigues(ocam[i, 4:7]) @ oRlw).max() < 1e-6
    f = ocam[:, 0]
    assert np.abs((disp[0] + disp[1] * f + disp[2] * f * f) - (od[0] + od[1] * f + od[2] * f * f)).max() < 1e-5


def test_ba_dist_disp_batch_and_shared_intrinsics(pkg, orc):
    """Scenes of a batch keep the bits of their solo solves (every scene its own displacement block); a non-zero initial
    block is honoured; SetSharedIntrinsics groups and the displacement group fold side by side."""
    scenes = [pkg.synth.make_scene(20 + i, 16 + 2 * i, 80, factor_type=3) for i in range(6)]
    b = pkg.api.BaBatch(scenes); b.set_state(); summ = b.solve(); cams, rays = b.get_state(); disps = b.get_disp(); b.close()
    for i in (0, 3, 5):
        cam1, ray1, s1, _, d1 = pkg.api.ba_solve_disp(scenes[i])
        assert s1 == summ[i]
        assert np.array_equal(cam1, cams[i]) and np.array_equal(ray1, rays[i]) and np.array_equal(d1, disps[i])
    # twelve scenes: the multi-launch factorisation and two scene groups
    more = scenes + [pkg.synth.make_scene(30 + i, 18, 80, factor_type=3) for i in range(6)]
    b = pkg.api.BaBatch(more); b.set_state(); summ12 = b.solve(); cams12, _ = b.get_state(); disps12 = b.get_disp(); b.close()
    for i in range(6):
        assert summ12[i] == summ[i] and np.array_equal(cams12[i], cams[i]) and np.array_equal(disps12[i], disps[i])
    # a start away from zero
    d0 = np.array([0.02, -1e-5, 2e-9])
    cam2, _, s2, _, d2 = pkg.api.ba_solve_disp(scenes[1], disp0=d0)
    od = d0.copy()
    ocam, _, _, osumm, _ = orc.ba_solve(scenes[1], jacobian_mode=orc.JAC_ANALYTIC, disp=od, num_threads=4)
    assert s2["termination_type"] == osumm["termination_type"] and s2["num_iterations"] == osumm["num_iterations"]
    assert abs(s2["initial_cost"] - osumm["initial_cost"]) / osumm["initial_cost"] < 1e-12
    assert _rel(cam2[:, 0], ocam[:, 0]) < 1e-6
    # shared intrinsics groups next to the displacement group
    sc = pkg.synth.make_scene(6, 24, 100, factor_type=3, n_intrinsics_groups=3)
    cam, _, s3, _, d3 = pkg.api.ba_solve_disp(sc)
    od = np.zeros(3)
    ocam, _, _, osumm, _ = orc.ba_solve(sc, jacobian_mode=orc.JAC_ANALYTIC, disp=od, num_threads=4)
    assert s3["termination_type"] == osumm["termination_type"] and s3["num_iterations"] == osumm["num_iterations"]
    # three focal lengths and a three-parameter delta(f): every group has a displacement of its own to trade against its focal
    # length, the valley is flat -- same LM path, cost to 1e-7 (1.2e-8 measured: where along the valley the last steps land
    # depends on the round-off of the Schur complement), focal lengths to 1e-4 only (1.3e-5 measured)
    assert abs(s3["final_cost"] - osumm["final_cost"]) / osumm["final_cost"] < 1e-7
    assert _rel(cam[:, 0], ocam[:, 0]) < 1e-4
    for g in np.unique(sc.ic_of_cam):
        m = np.flatnonzero(sc.ic_of_cam == g)
        assert np.all(cam[m, 0] == cam[m[0], 0]) and np.all(cam[m, 10] == cam[m[0], 10])


def test_disp_entry_points_reject_other_types(pkg):
    sc = pkg.synth.make_scene(0, 10, 60, factor_type=1)
    b = pkg.api.BaBatch([sc]); b.set_state()
    with pytest.raises(pkg.api.PtzError) as e:
        b.set_disp([np.zeros(3)])
    assert e.value.code == -4
    with pytest.raises(pkg.api.PtzError):
        b.get_disp()
    b.close()


def test_cpp_ptzray_optimizer_dist_disp(pkg, orc):
    """PTZRayOptimizer with FACTOR_TYPE PTZRayDistDisp through the C++ class: the packed problem solved by the same device
    path (identical summary to the C-ABI solve of the class's own packing), fy := fx and t_z += d0 + d1 fx + d2 fx^2 on
    read-back (ptzray_optimizer.cc:693, :714), 2D-2D error from PTZRayDistDispFactor residuals (:1010-1013)."""
    import host_util as hu
    from types import SimpleNamespace
    sc = pkg.synth.make_scene(5, 20, 100, factor_type=3)
    kps, plist = hu.scene_to_features_matches(sc)
    ok, cam, err, summ, pk = hu.ptzray_solve(kps, plist, sc.cam_init, max_iter=200, ftype=3)
    assert ok and summ["termination_type"] == 0
    base = dict(obs_uv=pk["obs_uv"], obs_cam=pk["obs_cam"], obs_ray=pk["obs_ray"], ray_weight=pk["ray_weight"], n_cam=sc.n_cam,
                n_ray=len(pk["ray_weight"]), factor_type=3)
    ns = SimpleNamespace(**base, cam_init=sc.cam_init, ray_init=orc.pix2ray(SimpleNamespace(**base), sc.cam_init))
    cam2, ray2, summ2, _, disp2 = pkg.api.ba_solve_disp(ns)
    # (the class initialises the rays itself; the oracle's Pix2Ray differs from it in the last bits)
    assert summ2["num_iterations"] == summ["num_iterations"] and abs(summ2["final_cost"] - summ["final_cost"]) < 1e-8 * summ["final_cost"]
    assert _rel(cam[:, 0], cam2[:, 0]) < 1e-7 and np.array_equal(cam[:, 1], cam[:, 0])
    delta = disp2[0] + disp2[1] * cam2[:, 0] + disp2[2] * cam2[:, 0] ** 2
    assert np.abs(delta).max() > 0 and np.allclose(cam[:, 9] - sc.cam_init[:, 9], delta, rtol=1e-5, atol=1e-9)
    # the C-ABI hands t_z and the displacement out separately; api.fold_displacement is the reference's last read-back step
    assert np.allclose(pkg.api.fold_displacement(cam2, disp2)[:, 9], cam[:, 9], rtol=1e-5, atol=1e-9)
    od = np.zeros(3)
    ocam, oray, _, osumm, _ = orc.ba_solve(ns, jacobian_mode=orc.JAC_ANALYTIC, disp=od, num_threads=4)
    assert summ["num_iterations"] == osumm["num_iterations"]
    assert _rel(cam[:, 0], ocam[:, 0]) < 1e-6
    res = orc.ba_residuals(ns, ocam, oray, disp=od)
    assert abs(err[1] - np.sqrt((res ** 2).sum() / len(res))) < 1e-6
