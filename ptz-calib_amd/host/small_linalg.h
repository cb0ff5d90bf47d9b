// small_linalg.h -- dense helpers for the host-side initialisation code (EPnP, homographies): one-sided Jacobi SVD
// and SVD-based least squares for matrices with at most a few dozen columns.  Row-major std::vector storage.
#pragma once

#include <vector>

namespace ptzcalib {

// Thin SVD A = U diag(s) V^T of an m x n matrix (m >= n), singular values in descending order.
// U: m x n (columns for zero singular values are zero), V: n x n.
void JacobiSVD(int m, int n, const std::vector<double>& A, std::vector<double>& U, std::vector<double>& s, std::vector<double>& V);
// The 3 x 3 case of JacobiSVD on arrays (no heap): the same rotations in the same order, hence the same bits -- what
// RodriguesInv (every Camera::ToVector) goes through.  U, V row-major 3 x 3, s descending.
void JacobiSVD3(const double* A_in, double* U, double* s, double* V);

// Minimum-norm least-squares solution of A x = b through the SVD (singular values below rcond * s_max dropped).
std::vector<double> SolveLeastSquares(int m, int n, const std::vector<double>& A, const std::vector<double>& b, double rcond = 1e-12);

// Eigen-decomposition of a symmetric positive semi-definite n x n matrix: eigenvalues descending, eigenvectors as
// the COLUMNS of V (row-major n x n).
void EigenSymPSD(int n, const std::vector<double>& A, std::vector<double>& evals, std::vector<double>& V);

}  // namespace ptzcalib
