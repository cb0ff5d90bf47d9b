"""Shared by the CPU and GPU tests that compare a solve with the independent minima of tests/golden/minima_*.json
(scipy.optimize.least_squares on residuals restated in tests/golden/gen_minima.py; nothing of oracle/ went into them)."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TIGHT = dict(function_tolerance=1e-14, parameter_tolerance=1e-14, gradient_tolerance=1e-14, max_num_iterations=200)


def rodrigues(rv):
    rv = np.asarray(rv, dtype=np.float64)
    th = np.linalg.norm(rv)
    if th < np.finfo(float).eps:
        return np.eye(3)
    k = rv / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * K


def load(name):
    return json.load(open(os.path.join(GOLD, f"minima_{name}.json")))


def scene_of(pkg, m):
    sid, nv, ob, ft = m["scene"]
    return pkg.synth.make_scene(sid, nv, ob, factor_type=ft)


def relrot(rvecs):
    R = [rodrigues(r) for r in rvecs]
    return np.stack([r @ R[0].T for r in R])


def check_against_minimum(cam, summ, m, tight):
    """cam: solved camera 15-vectors; m: the golden minimum.  Gauge-invariant quantities only (focal lengths, relative rotations,
    cost).  With Ceres' default function_tolerance 1e-6 a solve stops within ~1e-6 (relative) of the minimal cost, and its
    parameters at the distance that flat direction allows; with tight tolerances it reaches the minimum itself."""
    cmin = m["cost"]
    rel_cost = (summ["final_cost"] - cmin) / cmin
    f = np.asarray(m["focal"])
    df = np.abs(cam[:, 0] - f).max() / f.max()
    dr = np.abs(relrot(cam[:, 4:7]) - relrot(m["rvec"])).max()
    if tight:
        assert abs(rel_cost) < 1e-9, rel_cost
        assert df < 1e-6 and dr < 1e-6, (df, dr)
    else:
        assert -1e-9 < rel_cost < 5e-6, rel_cost   # never below the minimum; within what function_tolerance leaves
        assert df < 2e-3 and dr < 2e-3, (df, dr)
    if "k1" in m:
        assert np.abs(cam[:, 10] - np.asarray(m["k1"])).max() < (1e-6 if tight else 5e-3)
    return rel_cost, df, dr
