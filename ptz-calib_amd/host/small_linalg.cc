#include "small_linalg.h"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace ptzcalib {

// Hestenes one-sided Jacobi: rotate column pairs of A until they are mutually orthogonal; the column norms are the
// singular values, the accumulated rotations are V.
void JacobiSVD(int m, int n, const std::vector<double>& A_in, std::vector<double>& U, std::vector<double>& s, std::vector<double>& V)
{
  std::vector<double> A = A_in;
  std::vector<double> W(static_cast<size_t>(n) * n, 0.0);
  for (int j = 0; j < n; ++j) W[static_cast<size_t>(j) * n + j] = 1.0;
  const double eps = 1e-15;
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < m; ++i) {
          const double ap = A[static_cast<size_t>(i) * n + p], aq = A[static_cast<size_t>(i) * n + q];
          alpha += ap * ap; beta += aq * aq; gamma += ap * aq;
        }
        if (std::fabs(gamma) <= eps * std::sqrt(alpha * beta) || gamma == 0.0) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
        for (int i = 0; i < m; ++i) {
          double& ap = A[static_cast<size_t>(i) * n + p];
          double& aq = A[static_cast<size_t>(i) * n + q];
          const double x = ap, y = aq;
          ap = c * x - sn * y;
          aq = sn * x + c * y;
        }
        for (int i = 0; i < n; ++i) {
          double& vp = W[static_cast<size_t>(i) * n + p];
          double& vq = W[static_cast<size_t>(i) * n + q];
          const double x = vp, y = vq;
          vp = c * x - sn * y;
          vq = sn * x + c * y;
        }
      }
    if (!rotated) break;
  }
  std::vector<double> norm(n);
  for (int j = 0; j < n; ++j) {
    double a = 0;
    for (int i = 0; i < m; ++i) a += A[static_cast<size_t>(i) * n + j] * A[static_cast<size_t>(i) * n + j];
    norm[j] = std::sqrt(a);
  }
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return norm[a] > norm[b]; });
  U.assign(static_cast<size_t>(m) * n, 0.0);
  V.assign(static_cast<size_t>(n) * n, 0.0);
  s.assign(n, 0.0);
  for (int j = 0; j < n; ++j) {
    const int o = order[j];
    s[j] = norm[o];
    for (int i = 0; i < n; ++i) V[static_cast<size_t>(i) * n + j] = W[static_cast<size_t>(i) * n + o];
    if (norm[o] > 0)
      for (int i = 0; i < m; ++i) U[static_cast<size_t>(i) * n + j] = A[static_cast<size_t>(i) * n + o] / norm[o];
  }
}

void JacobiSVD3(const double* A_in, double* U, double* s, double* V)
{
  constexpr int m = 3, n = 3;
  double A[9], W[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 9; ++i) A[i] = A_in[i];
  const double eps = 1e-15;
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < m; ++i) {
          const double ap = A[i * n + p], aq = A[i * n + q];
          alpha += ap * ap; beta += aq * aq; gamma += ap * aq;
        }
        if (std::fabs(gamma) <= eps * std::sqrt(alpha * beta) || gamma == 0.0) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
        for (int i = 0; i < m; ++i) {
          const double x = A[i * n + p], y = A[i * n + q];
          A[i * n + p] = c * x - sn * y;
          A[i * n + q] = sn * x + c * y;
        }
        for (int i = 0; i < n; ++i) {
          const double x = W[i * n + p], y = W[i * n + q];
          W[i * n + p] = c * x - sn * y;
          W[i * n + q] = sn * x + c * y;
        }
      }
    if (!rotated) break;
  }
  double norm[3];
  for (int j = 0; j < n; ++j) {
    double a = 0;
    for (int i = 0; i < m; ++i) a += A[i * n + j] * A[i * n + j];
    norm[j] = std::sqrt(a);
  }
  int order[3] = {0, 1, 2};
  // stable descending sort of three (insertion: equal norms keep their order, as std::stable_sort does)
  for (int a = 1; a < 3; ++a)
    for (int b2 = a; b2 > 0 && norm[order[b2]] > norm[order[b2 - 1]]; --b2) std::swap(order[b2], order[b2 - 1]);
  for (int i = 0; i < 9; ++i) { U[i] = 0.0; V[i] = 0.0; }
  for (int j = 0; j < n; ++j) {
    const int o = order[j];
    s[j] = norm[o];
    for (int i = 0; i < n; ++i) V[i * n + j] = W[i * n + o];
    if (norm[o] > 0)
      for (int i = 0; i < m; ++i) U[i * n + j] = A[i * n + o] / norm[o];
  }
}

std::vector<double> SolveLeastSquares(int m, int n, const std::vector<double>& A, const std::vector<double>& b, double rcond)
{
  std::vector<double> U, s, V;
  JacobiSVD(m, n, A, U, s, V);
  std::vector<double> x(n, 0.0);
  for (int j = 0; j < n; ++j) {
    if (!(s[j] > rcond * s[0])) continue;
    double ub = 0;
    for (int i = 0; i < m; ++i) ub += U[static_cast<size_t>(i) * n + j] * b[i];
    ub /= s[j];
    for (int i = 0; i < n; ++i) x[i] += V[static_cast<size_t>(i) * n + j] * ub;
  }
  return x;
}

void EigenSymPSD(int n, const std::vector<double>& A, std::vector<double>& evals, std::vector<double>& V)
{
  std::vector<double> U;
  JacobiSVD(n, n, A, U, evals, V);  // symmetric PSD: singular values = eigenvalues, right singular vectors = eigenvectors
}

}  // namespace ptzcalib
