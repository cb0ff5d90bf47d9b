"""Probe (not a test): dense solver alone -- accuracy against LAPACK and device time of factor + solve for one / a few systems."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
import ctypes as C
lib = pkg.api.lib()
def run(count, n, reps=5):
    rng = np.random.default_rng(n)
    M = rng.standard_normal((count, n, n + 8))
    A = M @ np.transpose(M, (0, 2, 1)) + 1e-3 * np.eye(n)
    rhs = rng.standard_normal((count, n))
    best = 1e9
    for _ in range(reps):
        x = np.zeros((count, n)); fail = np.zeros(count, dtype=np.int32); ms = C.c_double()
        rc = lib.ptz_chol_solve_batch(count, n, A.ctypes.data_as(C.c_void_p), rhs.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p),
                                      fail.ctypes.data_as(C.c_void_p), 0, C.byref(ms))
        assert rc == 0
        best = min(best, ms.value)
    err = max(np.abs(x[i] - np.linalg.solve(A[i], rhs[i])).max() / np.abs(x[i]).max() for i in range(count))
    print(f"count {count} n {n}: {best * 1e3:.1f} us, max rel err vs LAPACK {err:.2e}, fail {fail.sum()}", flush=True)
for count, n in [(1, 63), (1, 200), (1, 800), (4, 800), (1, 1600)]:
    run(count, n)
