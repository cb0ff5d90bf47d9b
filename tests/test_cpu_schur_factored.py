"""The algebra of k_schur_f (ptz-calib_amd/csrc/ptz_ba_kernels.h), restated in numpy and held to the direct product it replaces.

For PTZRay (pinhole, no distortion; ptzray_optimizer.cc:20-56) the off-diagonal block of the Schur complement that two
observations a (camera i) and b (camera j) of ONE ray contribute is  W_a E W_b^T  with  W = Jc^T Jr  (weighted, Jacobi-scaled) and
E = (V + D^2)^-1 of the ray.  The kernel never forms W: it keeps, per observation a of camera i, the 2 x 3 matrix
Q_a = Pz_i^-1 [Jr0_a E''] R_i^T and the normalised image point (x_a, y_a) -- 8 doubles -- and an entry is
    Graw_a^T [P'z^-1 Q_a (Rji(0:1) - (x_b, y_b) Rji(2))^T] Graw_b,     P' = R_j R_i^T (x_a, y_a, 1),
with the factors that are constant over the camera pair applied to the pair's sum.  This test follows the kernel's steps literally
(k_ray_prep's folding, phase 1's row, phase 2's entry, phase 3's transforms) and compares with W_a E W_b^T computed from the
functor's Jacobians.  No GPU, no library: it pins the derivation, the GPU parity tests pin the implementation."""
import numpy as np


def rodrigues(r):
    t = np.linalg.norm(r)
    if t < 2.220446049250313e-16:
        return np.eye(3)
    k = r / t
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(t) * np.eye(3) + (1 - np.cos(t)) * np.outer(k, k) + np.sin(t) * K


def so3_left_jacobian(r):
    t2 = r @ r
    t = np.sqrt(t2)
    a = 2.0 * np.sin(0.5 * t) ** 2 / t2
    b = (t - np.sin(t)) / (t2 * t)
    Kx = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    return np.eye(3) + a * Kx + b * (Kx @ Kx)


def jacobians(f, R, Jl, X):
    """Unweighted, unscaled Jacobians of the PTZRay residual uv - (f x + cx, f y + cy) wrt [f, r1, r2, r3] and wrt the ray X."""
    n = np.linalg.norm(X)
    Xn = X / n
    P = R @ Xn
    x, y = P[0] / P[2], P[1] / P[2]
    M = f / P[2] * np.array([[1, 0, -x], [0, 1, -y]])
    Jc = np.zeros((2, 4))
    Jc[:, 0] = [-x, -y]
    for k in range(3):
        Jc[:, 1 + k] = -M @ np.cross(Jl[:, k], P)
    Jr = -(M @ R) / n
    return Jc, Jr, Xn


def graw(x, y):
    return np.array([[-x, x * y, -(1 + x * x), y], [-y, 1 + y * y, -x * y, -x]])


def test_factored_entry_equals_the_direct_product():
    rng = np.random.default_rng(11)
    worst = 0.0
    for trial in range(200):
        ri, rj = rng.normal(size=3) * 0.4, rng.normal(size=3) * 0.4
        Ri, Rj = rodrigues(ri), rodrigues(rj)
        Jli, Jlj = so3_left_jacobian(ri), so3_left_jacobian(rj)
        fi, fj = rng.uniform(1500, 3500, 2)
        si, sj = rng.uniform(1e-3, 1.0, 4), rng.uniform(1e-3, 1.0, 4)          # Jacobi scales of the camera columns
        sr = rng.uniform(1e-2, 1.0, 3)                                           # ... of the ray
        w = float(rng.integers(2, 12))                                           # ScaledLoss(track length)
        X = Ri.T @ np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2), 1.0]) * rng.uniform(0.5, 3.0)
        A = rng.normal(size=(3, 3)); E = np.linalg.inv(A @ A.T + np.eye(3))      # any SPD (V + D^2)^-1
        sw = np.sqrt(w)
        # ---- the direct product
        Jci, Jri, Xn = jacobians(fi, Ri, Jli, X)
        Jcj, Jrj, _ = jacobians(fj, Rj, Jlj, X)
        Wa = (sw * Jci * si).T @ (sw * Jri * sr)
        Wb = (sw * Jcj * sj).T @ (sw * Jrj * sr)
        want = Wa @ E @ Wb.T
        # ---- k_ray_prep (Dev::e_fold): a = sqrt(w) s_r / |X|, c = sqrt(w) a, E'' = diag(c) E diag(c)
        inv_n = 1.0 / np.linalg.norm(X)
        al = sw * inv_n * sr
        c = sw * al
        Epp = (c[:, None] * E) * c[None, :]
        # ---- phase 1, observation a of camera i: the row {Q_a, x_a, y_a}
        P = Ri @ Xn
        iz = 1.0 / P[2]
        xa, ya = P[0] * iz, P[1] * iz
        fiz = fi * iz
        Jr0 = -fiz * np.array([Ri[0] - xa * Ri[2], Ri[1] - ya * Ri[2]])
        Q = iz * (Jr0 @ Epp) @ Ri.T
        # ---- phase 2, the entry with camera j
        Rji = Rj @ Ri.T
        Pp = Rji @ np.array([xa, ya, 1.0])
        izp = 1.0 / Pp[2]
        xb, yb = Pp[0] * izp, Pp[1] * izp
        Mr = np.array([Rji[0] - xb * Rji[2], Rji[1] - yb * Rji[2]])
        K = izp * (Q @ Mr.T)
        acc = graw(xa, ya).T @ K @ graw(xb, yb)
        # ---- phase 3: rows through (F_i B_i)^T, columns through f_j F_j B_j.  What the kernel stores is the block of the Schur
        #      complement S = U - W E W^T, i.e. S_ij = -W_a E W_b^T summed over the pair's entries: the row carries Jr0_a = -MR_a
        #      with its sign, the entry uses +MR_b, so the transformed sum IS -W_a E W_b^T and is stored as it is
        Fi = np.diag([1.0, fi, fi, fi]); Fj = np.diag([1.0, fj, fj, fj])
        Bi = np.block([[np.ones((1, 1)), np.zeros((1, 3))], [np.zeros((3, 1)), Jli]]) @ np.diag(si)
        Bj = np.block([[np.ones((1, 1)), np.zeros((1, 3))], [np.zeros((3, 1)), Jlj]]) @ np.diag(sj)
        got = (Fi @ Bi).T @ acc @ (fj * Fj @ Bj)
        err = np.abs(got + want).max() / np.abs(want).max()
        worst = max(worst, err)
        assert err < 1e-11, (trial, err)
    assert worst < 1e-11


def test_factored_diagonal_sums_equal_the_direct_products():
    """Phase 1's other two results, per observation: the diagonal block's share W_a E W_a^T = Jc0^T (Jr0 E'' Jr0^T) Jc0 and the
    right-hand side's share W_a z = Jc0^T (Jr0 z''), with the weight and the ray-side factors living in E'' and z''."""
    rng = np.random.default_rng(5)
    for trial in range(100):
        r = rng.normal(size=3) * 0.5
        R, Jl = rodrigues(r), so3_left_jacobian(r)
        f = rng.uniform(1500, 3500)
        s, sr = rng.uniform(1e-3, 1.0, 4), rng.uniform(1e-2, 1.0, 3)
        w = float(rng.integers(2, 12)); sw = np.sqrt(w)
        X = R.T @ np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2), 1.0]) * rng.uniform(0.5, 3.0)
        A = rng.normal(size=(3, 3)); E = np.linalg.inv(A @ A.T + np.eye(3))
        g = rng.normal(size=3); z = E @ g
        Jc, Jr, Xn = jacobians(f, R, Jl, X)
        W = (sw * Jc * s).T @ (sw * Jr * sr)
        want_D, want_b = W @ E @ W.T, W @ z
        al = sw / np.linalg.norm(X) * sr; c = sw * al
        Epp = (c[:, None] * E) * c[None, :]; zpp = c * z
        P = R @ Xn; iz = 1.0 / P[2]; x, y = P[0] * iz, P[1] * iz
        Jr0 = -(f * iz) * np.array([R[0] - x * R[2], R[1] - y * R[2]])
        G = graw(x, y) @ np.diag([1.0, f, f, f])
        B = np.block([[np.ones((1, 1)), np.zeros((1, 3))], [np.zeros((3, 1)), Jl]]) @ np.diag(s)
        Jc0 = G @ B
        got_D = Jc0.T @ (Jr0 @ Epp @ Jr0.T) @ Jc0
        got_b = Jc0.T @ (Jr0 @ zpp)
        assert np.abs(got_D - want_D).max() / np.abs(want_D).max() < 1e-11
        assert np.abs(got_b - want_b).max() / np.abs(want_b).max() < 1e-11
