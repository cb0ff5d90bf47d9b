#!/usr/bin/env python3
"""Independent minima for the parity tests (build container only; the output JSON files are the committed fixtures).

The floating-point hot path of the reference cannot be pinned against the reference itself (Ceres / OpenCV are absent, no
golden vectors exist).  What CAN be pinned independently of oracle/ is WHERE the minimum of the reference's objective lies:
this script restates the residual definitions in numpy from the reference's functors --

  PTZRayFactor      src/core/ptzray_optimizer.cc:20-56    r = uv - (f pi(R(rvec) X/|X|) + c)
  PTZRayDistFactor  src/core/ptzray_optimizer.cc:65-129   r = uv - (f Brown(pi(R(rvec) X); k1) + c)      (no normalisation of X)
  ScaledLoss        src/core/ptzray_optimizer.cc:805-806  rows scaled by sqrt(track length)
  Factor2d2d        src/core/krt_optimizer.cc:22-43       r = uv2 - (f pi(R(rvec) normalise(K1^-1 [uv1, 1])) + c2)

-- and minimises them with scipy.optimize.least_squares (trust-region reflective, x_scale='jac', sparse finite-difference
Jacobian, tolerances at machine precision), i.e. with a solver that shares no code and no algorithmic choice with oracle/ or with
the device path.  Nothing under oracle/ is imported.  usage: python tests/golden/gen_minima.py
"""
import json
import os
import sys
import time

import numpy as np
from scipy.optimize import least_squares
from scipy.sparse import lil_matrix

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()  # the synthetic scene generator only (inputs); no solver code is used


def rodrigues_batch(rv):
    th = np.linalg.norm(rv, axis=1)
    small = th < np.finfo(float).eps
    ths = np.where(small, 1.0, th)
    k = rv / ths[:, None]
    c, s = np.cos(th), np.sin(th)
    K = np.zeros((len(rv), 3, 3))
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -k[:, 2], k[:, 1], k[:, 2], -k[:, 0], -k[:, 1], k[:, 0]
    R = c[:, None, None] * np.eye(3)[None] + (1 - c)[:, None, None] * k[:, :, None] * k[:, None, :] + s[:, None, None] * K
    R[small] = np.eye(3)
    return R


def ba_problem(sc):
    """Parameter vector: [f (N) | k1 (N, Dist only) | rvec (3N) | X (3P)]; returns (x0, fun, sparsity)."""
    N, P, dist = sc.n_cam, sc.n_ray, sc.factor_type == 1
    cx, cy = sc.cam_init[0, 2], sc.cam_init[0, 3]
    oc, orr = sc.obs_cam.astype(int), sc.obs_ray.astype(int)
    uv = sc.obs_uv.astype(np.float64)
    sw = np.sqrt(sc.ray_weight)[orr]
    nk = N if dist else 0

    def unpack(x):
        f = x[:N]
        k1 = x[N:N + nk] if dist else None
        rv = x[N + nk:N + nk + 3 * N].reshape(N, 3)
        X = x[N + nk + 3 * N:].reshape(P, 3)
        return f, k1, rv, X

    def fun(x):
        f, k1, rv, X = unpack(x)
        R = rodrigues_batch(rv)
        Xo = X[orr] if dist else X[orr] / np.linalg.norm(X[orr], axis=1, keepdims=True)
        Pc = np.einsum("nij,nj->ni", R[oc], Xo)
        xn, yn = Pc[:, 0] / Pc[:, 2], Pc[:, 1] / Pc[:, 2]
        if dist:
            r2 = xn * xn + yn * yn
            rad = 1.0 + k1[oc] * r2
            xn, yn = xn * rad, yn * rad
        ru = (uv[:, 0] - (f[oc] * xn + cx)) * sw
        rvv = (uv[:, 1] - (f[oc] * yn + cy)) * sw
        return np.stack([ru, rvv], axis=1).ravel()

    x0 = np.concatenate([sc.cam_init[:, 0]] + ([sc.cam_init[:, 10]] if dist else []) + [sc.cam_init[:, 4:7].ravel(), sc.ray_init.ravel()])
    nobs = len(oc)
    S = lil_matrix((2 * nobs, len(x0)), dtype=np.int8)
    rows = np.arange(nobs)
    for comp in (0, 1):
        r = 2 * rows + comp
        S[r, oc] = 1
        if dist:
            S[r, N + oc] = 1
        for k in range(3):
            S[r, N + nk + 3 * oc + k] = 1
            S[r, N + nk + 3 * N + 3 * orr + k] = 1
    return x0, fun, S.tocsr(), unpack


def solve_ba(sc, name):
    x0, fun, S, unpack = ba_problem(sc)
    t = time.time()
    res = least_squares(fun, x0, jac="2-point", jac_sparsity=S, method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15,
                        max_nfev=400, tr_solver="lsmr", tr_options={"atol": 1e-14, "btol": 1e-14, "maxiter": 4000})
    # polish: a few Gauss-Newton steps of the same solver from its own end point, central differences
    res = least_squares(fun, res.x, jac="3-point", jac_sparsity=S, method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15,
                        max_nfev=100, tr_solver="lsmr", tr_options={"atol": 1e-15, "btol": 1e-15, "maxiter": 8000})
    f, k1, rv, X = unpack(res.x)
    g = res.jac.T @ res.fun
    print(f"{name}: cost {0.5 * np.dot(res.fun, res.fun):.12f} (initial {0.5 * np.dot(fun(x0), fun(x0)):.3f}), |grad|_inf {np.abs(g).max():.2e}, "
          f"nfev {res.nfev}, {time.time() - t:.1f} s, status {res.status}")
    out = {"cost": 0.5 * float(np.dot(res.fun, res.fun)), "focal": f.tolist(), "rvec": rv.tolist(), "ray": X.tolist(), "grad_inf": float(np.abs(g).max()),
           "solver": "scipy.optimize.least_squares(trf, x_scale='jac', lsmr), residuals restated in tests/golden/gen_minima.py"}
    if k1 is not None:
        out["k1"] = k1.tolist()
    return out


def solve_reloc(rb, q, dist):
    """One relocalization query in the reference camera's local frame (krt_optimizer.cc:265-348): parameters [f, rvec (, k1)]."""
    s = slice(rb.match_ptr[q], rb.match_ptr[q + 1])
    cr, ci = rb.cam_ref[q], rb.cam_init[q]
    u1 = rb.uv_ref[s].astype(np.float64); u2 = rb.uv_cur[s].astype(np.float64)
    Rr = rodrigues_batch(cr[None, 4:7])[0]
    Ri = rodrigues_batch(ci[None, 4:7])[0]
    ray1 = np.stack([(u1[:, 0] - cr[2]) / cr[0], (u1[:, 1] - cr[3]) / cr[1], np.ones(len(u1))], axis=1)
    ray1 /= np.linalg.norm(ray1, axis=1, keepdims=True)
    Rloc0 = Ri @ Rr.T  # current camera relative to the reference camera

    def rvec_of(Rm):
        return pkg.synth.rodrigues_inv(Rm)

    def fun(x):
        R = rodrigues_batch(x[None, 1:4])[0]
        Pc = ray1 @ R.T
        xn, yn = Pc[:, 0] / Pc[:, 2], Pc[:, 1] / Pc[:, 2]
        if dist:
            r2 = xn * xn + yn * yn
            rad = 1.0 + x[4] * r2
            xn, yn = xn * rad, yn * rad
        return np.stack([u2[:, 0] - (x[0] * xn + ci[2]), u2[:, 1] - (x[0] * yn + ci[3])], axis=1).ravel()

    x0 = np.concatenate([[ci[0]], rvec_of(Rloc0)] + ([[ci[10]]] if dist else []))
    res = least_squares(fun, x0, jac="3-point", method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=200)
    Rw = rodrigues_batch(res.x[None, 1:4])[0] @ Rr  # back to the world frame
    return {"cost": 0.5 * float(np.dot(res.fun, res.fun)), "focal": float(res.x[0]), "R_world": Rw.tolist(),
            **({"k1": float(res.x[4])} if dist else {})}


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    cases = {"c1": pkg.synth.make_scene(0, 20, 100), "c1_dist": pkg.synth.make_scene(3, 20, 100, factor_type=1),
             "m60x300": pkg.synth.make_scene(5, 60, 300)}
    for name, sc in cases.items():
        res = solve_ba(sc, name)
        res["scene"] = {"c1": [0, 20, 100, 0], "c1_dist": [3, 20, 100, 1], "m60x300": [5, 60, 300, 0]}[name]
        json.dump(res, open(os.path.join(out_dir, f"minima_{name}.json"), "w"))
    rel = {}
    for ft in (0, 1):
        rb = pkg.synth.make_reloc_batch(16, 128, seed_id=ft, factor_type=ft)
        rel[str(ft)] = [solve_reloc(rb, q, ft == 1) for q in range(rb.n_query)]
    json.dump({"batch": "synth.make_reloc_batch(16, 128, seed_id=ft, factor_type=ft)", "queries": rel}, open(os.path.join(out_dir, "minima_reloc.json"), "w"))


if __name__ == "__main__":
    main()
