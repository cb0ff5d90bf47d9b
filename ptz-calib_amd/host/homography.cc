#include "homography.h"

#include <algorithm>
#include <cmath>
#include <cstdint>

#include "small_linalg.h"

namespace ptzcalib {
namespace {

struct Rng {  // SplitMix64: deterministic across platforms
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t Next()
  {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  int Below(int n) { return static_cast<int>(Next() % static_cast<uint64_t>(n)); }
};

void Normalisation(const std::vector<Point2f>& p, const std::vector<int>& idx, double& cx, double& cy, double& s)
{
  cx = cy = 0;
  for (int i : idx) { cx += p[i].x; cy += p[i].y; }
  cx /= idx.size(); cy /= idx.size();
  double d = 0;
  for (int i : idx) d += std::sqrt((p[i].x - cx) * (p[i].x - cx) + (p[i].y - cy) * (p[i].y - cy));
  d /= idx.size();
  s = d > 1e-12 ? std::sqrt(2.0) / d : 1.0;
}

double TransferError2(const Mat33& H, const Point2f& a, const Point2f& b)
{
  const double w = H[6] * a.x + H[7] * a.y + H[8];
  const double iw = std::fabs(w) > 2.2e-16 ? 1.0 / w : 0.0;  // a point mapped to infinity counts as a gross error
  const double dx = (H[0] * a.x + H[1] * a.y + H[2]) * iw - b.x, dy = (H[3] * a.x + H[4] * a.y + H[5]) * iw - b.y;
  return dx * dx + dy * dy;
}

// three of the four sample points (nearly) on a line make the minimal problem degenerate
bool DegenerateSample(const std::vector<Point2f>& p, const int s[4])
{
  for (int a = 0; a < 4; ++a)
    for (int b = a + 1; b < 4; ++b)
      for (int c = b + 1; c < 4; ++c) {
        const double x1 = p[s[b]].x - p[s[a]].x, y1 = p[s[b]].y - p[s[a]].y, x2 = p[s[c]].x - p[s[a]].x, y2 = p[s[c]].y - p[s[a]].y;
        if (std::fabs(x1 * y2 - x2 * y1) <= 1e-7 * (std::fabs(x1) + std::fabs(y1) + std::fabs(x2) + std::fabs(y2))) return true;
      }
  return false;
}

// Gauss-Newton on the forward transfer error over the 8 free entries (h33 = 1)
void Refine(const std::vector<Point2f>& src, const std::vector<Point2f>& dst, const std::vector<int>& idx, Mat33& H)
{
  auto cost = [&](const Mat33& M) { double c = 0; for (int i : idx) c += TransferError2(M, src[i], dst[i]); return c; };
  double cur = cost(H);
  double lambda = 1e-6;
  for (int it = 0; it < 10; ++it) {
    std::vector<double> JtJ(64, 0.0), Jtr(8, 0.0);
    for (int i : idx) {
      const double x = src[i].x, y = src[i].y;
      const double w = H[6] * x + H[7] * y + H[8], iw = 1.0 / w;
      const double u = (H[0] * x + H[1] * y + H[2]) * iw, v = (H[3] * x + H[4] * y + H[5]) * iw;
      const double ju[8] = {x * iw, y * iw, iw, 0, 0, 0, -u * x * iw, -u * y * iw};
      const double jv[8] = {0, 0, 0, x * iw, y * iw, iw, -v * x * iw, -v * y * iw};
      const double ru = u - dst[i].x, rv = v - dst[i].y;
      for (int a = 0; a < 8; ++a) {
        Jtr[a] += ju[a] * ru + jv[a] * rv;
        for (int b = 0; b < 8; ++b) JtJ[8 * a + b] += ju[a] * ju[b] + jv[a] * jv[b];
      }
    }
    bool improved = false;
    for (int attempt = 0; attempt < 6 && !improved; ++attempt) {
      std::vector<double> A = JtJ, b(8);
      for (int a = 0; a < 8; ++a) { A[8 * a + a] *= 1.0 + lambda; b[a] = -Jtr[a]; }
      const std::vector<double> d = SolveLeastSquares(8, 8, A, b);
      Mat33 Hn = H;
      for (int a = 0; a < 8; ++a) Hn[a] += d[a];
      const double c = cost(Hn);
      if (c < cur) {
        const bool tiny = cur - c <= 1e-12 * cur;
        H = Hn; cur = c; lambda = std::max(lambda * 0.1, 1e-12); improved = true;
        if (tiny) return;
      }
      else lambda *= 10;
    }
    if (!improved) return;
  }
}

}  // namespace

bool FitHomographyDLT(const std::vector<Point2f>& src, const std::vector<Point2f>& dst, const std::vector<int>& idx, Mat33& H)
{
  if (idx.size() < 4) return false;
  double cxa, cya, sa, cxb, cyb, sb;
  Normalisation(src, idx, cxa, cya, sa);
  Normalisation(dst, idx, cxb, cyb, sb);
  std::vector<double> AtA(81, 0.0);
  for (int i : idx) {
    const double x = (src[i].x - cxa) * sa, y = (src[i].y - cya) * sa, u = (dst[i].x - cxb) * sb, v = (dst[i].y - cyb) * sb;
    const double r0[9] = {x, y, 1, 0, 0, 0, -u * x, -u * y, -u};
    const double r1[9] = {0, 0, 0, x, y, 1, -v * x, -v * y, -v};
    for (int a = 0; a < 9; ++a)
      for (int b = 0; b < 9; ++b) AtA[9 * a + b] += r0[a] * r0[b] + r1[a] * r1[b];
  }
  std::vector<double> ev, V;
  EigenSymPSD(9, AtA, ev, V);
  Mat33 Hn;
  for (int k = 0; k < 9; ++k) Hn[k] = V[9 * k + 8];
  // undo the normalisations: H = Tb^-1 Hn Ta
  const Mat33 Ta = {sa, 0, -sa * cxa, 0, sa, -sa * cya, 0, 0, 1};
  const Mat33 Tbi = {1.0 / sb, 0, cxb, 0, 1.0 / sb, cyb, 0, 0, 1};
  H = Mul(Mul(Tbi, Hn), Ta);
  if (!(std::fabs(H[8]) > 1e-300)) return false;
  const double inv = 1.0 / H[8];
  for (double& h : H) h *= inv;
  for (double h : H) if (!std::isfinite(h)) return false;
  return true;
}

bool FindHomographyRansac(const std::vector<Point2f>& src, const std::vector<Point2f>& dst, double ransac_thresh, Mat33& H_out,
                          std::vector<unsigned char>* inlier_mask)
{
  const int n = static_cast<int>(src.size());
  if (n < 4 || dst.size() != src.size()) return false;
  const double thr2 = ransac_thresh * ransac_thresh;
  std::vector<int> all(n);
  for (int i = 0; i < n; ++i) all[i] = i;
  Mat33 best = Eye3();
  int best_inliers = 0;
  if (n == 4) {
    if (!FitHomographyDLT(src, dst, all, best)) return false;
    best_inliers = 4;
  }
  else {
    Rng rng(0x50545A48u);
    const double confidence = 0.995;
    int max_iters = 2000;
    for (int it = 0; it < max_iters; ++it) {
      int s[4];
      int tries = 0;
      bool ok = false;
      for (; tries < 100 && !ok; ++tries) {
        for (int k = 0; k < 4;) {
          s[k] = rng.Below(n);
          bool dup = false;
          for (int q = 0; q < k; ++q) dup |= (s[q] == s[k]);
          if (!dup) ++k;
        }
        ok = !DegenerateSample(src, s) && !DegenerateSample(dst, s);
      }
      if (!ok) continue;
      Mat33 Hs;
      if (!FitHomographyDLT(src, dst, {s[0], s[1], s[2], s[3]}, Hs)) continue;
      int cnt = 0;
      for (int i = 0; i < n; ++i) cnt += TransferError2(Hs, src[i], dst[i]) <= thr2;
      if (cnt > std::max(best_inliers, 3)) {
        best_inliers = cnt;
        best = Hs;
        // iterations needed to draw one outlier-free sample with the requested confidence
        const double ep = 1.0 - static_cast<double>(cnt) / n;
        const double denom = std::log(std::max(1.0 - std::pow(1.0 - ep, 4), 1e-300));
        const double need = (denom >= 0 || ep <= 0) ? 0 : std::log(1.0 - confidence) / denom;
        max_iters = std::min(max_iters, std::max(it + 1, static_cast<int>(std::ceil(need))));
      }
    }
    if (best_inliers < 4) return false;
  }
  std::vector<int> inl;
  for (int i = 0; i < n; ++i) if (TransferError2(best, src[i], dst[i]) <= thr2) inl.push_back(i);
  if (inl.size() < 4) return false;
  Mat33 H = best;
  if (FitHomographyDLT(src, dst, inl, H)) Refine(src, dst, inl, H);
  else H = best;
  const double inv = 1.0 / H[8];
  for (double& h : H) h *= inv;
  if (inlier_mask) {
    inlier_mask->assign(n, 0);
    for (int i = 0; i < n; ++i) (*inlier_mask)[i] = TransferError2(H, src[i], dst[i]) <= thr2;
  }
  H_out = H;
  return true;
}

}  // namespace ptzcalib
