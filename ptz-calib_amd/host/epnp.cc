#include "epnp.h"

#include <algorithm>
#include <cmath>
#include <limits>

#include "small_linalg.h"

namespace ptzcalib {
namespace {

struct Problem {
  int n = 0;
  std::vector<double> pw;  // [3n] world points
  std::vector<double> uv;  // [2n] normalised, undistorted image points
};

double Det3(const Mat33& m)
{
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// nearest rotation to a 3x3 matrix (polar factor through the SVD, determinant forced to +1)
Mat33 NearestRotation(const Mat33& A)
{
  std::vector<double> U, s, V;
  JacobiSVD(3, 3, std::vector<double>(A.begin(), A.end()), U, s, V);
  // a zero singular value leaves a zero column in U: complete it with the cross product of the other two
  if (!(s[2] > 1e-12 * s[0])) {
    U[2] = U[3] * U[7] - U[6] * U[4];
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
  }
  Mat33 Um, Vm;
  for (int i = 0; i < 9; ++i) { Um[i] = U[i]; Vm[i] = V[i]; }
  Mat33 R = Mul(Um, Transpose(Vm));
  if (Det3(R) < 0) {
    for (int i = 0; i < 3; ++i) Um[3 * i + 2] = -Um[3 * i + 2];
    R = Mul(Um, Transpose(Vm));
  }
  return R;
}

// mean reprojection distance in normalised coordinates; +inf when a point falls behind the camera plane
double ReprojError(const Problem& p, const Mat33& R, const Vec3& t)
{
  double sum = 0;
  for (int i = 0; i < p.n; ++i) {
    const Vec3 X = Mul(R, Vec3{p.pw[3 * i], p.pw[3 * i + 1], p.pw[3 * i + 2]});
    const double z = X[2] + t[2];
    if (!(std::fabs(z) > 0)) return std::numeric_limits<double>::infinity();
    const double du = (X[0] + t[0]) / z - p.uv[2 * i], dv = (X[1] + t[1]) / z - p.uv[2 * i + 1];
    sum += std::sqrt(du * du + dv * dv);
  }
  return sum / p.n;
}

// Rigid alignment of the world points onto their camera-frame estimates (Arun / Horn through the SVD).
void AbsoluteOrientation(const Problem& p, const std::vector<double>& pc, Mat33& R, Vec3& t)
{
  Vec3 cw = {0, 0, 0}, cc = {0, 0, 0};
  for (int i = 0; i < p.n; ++i)
    for (int k = 0; k < 3; ++k) { cw[k] += p.pw[3 * i + k]; cc[k] += pc[3 * i + k]; }
  for (int k = 0; k < 3; ++k) { cw[k] /= p.n; cc[k] /= p.n; }
  Mat33 S = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < p.n; ++i)
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) S[3 * r + c] += (pc[3 * i + r] - cc[r]) * (p.pw[3 * i + c] - cw[c]);
  R = NearestRotation(S);
  const Vec3 Rc = Mul(R, cw);
  t = {cc[0] - Rc[0], cc[1] - Rc[1], cc[2] - Rc[2]};
}

// Damped Gauss-Newton on the six pose parameters (left-multiplicative rotation update), minimising the reprojection
// error in normalised coordinates.  The algebraic solutions above are exact for exact data but not statistically
// optimal; a few steps remove most of the noise amplification of distant, narrow-field configurations.
void PolishPose(const Problem& p, Mat33& R, Vec3& t)
{
  auto cost = [&](const Mat33& Rc, const Vec3& tc) {
    double c = 0;
    for (int i = 0; i < p.n; ++i) {
      const Vec3 X = Mul(Rc, Vec3{p.pw[3 * i], p.pw[3 * i + 1], p.pw[3 * i + 2]});
      const double z = X[2] + tc[2];
      if (!(z > 0)) return std::numeric_limits<double>::infinity();
      const double du = (X[0] + tc[0]) / z - p.uv[2 * i], dv = (X[1] + tc[1]) / z - p.uv[2 * i + 1];
      c += du * du + dv * dv;
    }
    return c;
  };
  double cur = cost(R, t);
  if (!std::isfinite(cur)) return;
  double lambda = 1e-4;
  for (int it = 0; it < 20; ++it) {
    double H[36] = {0}, g[6] = {0};
    for (int i = 0; i < p.n; ++i) {
      const Vec3 Y = Mul(R, Vec3{p.pw[3 * i], p.pw[3 * i + 1], p.pw[3 * i + 2]});
      const double X = Y[0] + t[0], Yc = Y[1] + t[1], Z = Y[2] + t[2];
      const double iz = 1.0 / Z, u = X * iz, v = Yc * iz;
      // d(u,v)/dP, P = exp(w) Y + t  ->  dP/dw = -[Y]x, dP/dt = I
      const double Ju[3] = {iz, 0, -u * iz}, Jv[3] = {0, iz, -v * iz};
      double ju[6], jv[6];
      ju[0] = Ju[1] * (-Y[2]) + Ju[2] * Y[1]; ju[1] = Ju[0] * Y[2] + Ju[2] * (-Y[0]); ju[2] = Ju[0] * (-Y[1]) + Ju[1] * Y[0];
      jv[0] = Jv[1] * (-Y[2]) + Jv[2] * Y[1]; jv[1] = Jv[0] * Y[2] + Jv[2] * (-Y[0]); jv[2] = Jv[0] * (-Y[1]) + Jv[1] * Y[0];
      for (int k = 0; k < 3; ++k) { ju[3 + k] = Ju[k]; jv[3 + k] = Jv[k]; }
      const double ru = u - p.uv[2 * i], rv = v - p.uv[2 * i + 1];
      for (int a = 0; a < 6; ++a) {
        g[a] += ju[a] * ru + jv[a] * rv;
        for (int b = 0; b < 6; ++b) H[6 * a + b] += ju[a] * ju[b] + jv[a] * jv[b];
      }
    }
    bool improved = false;
    for (int attempt = 0; attempt < 8 && !improved; ++attempt) {
      std::vector<double> A(H, H + 36), b(6);
      for (int a = 0; a < 6; ++a) { A[6 * a + a] += lambda * (H[6 * a + a] + 1e-12); b[a] = -g[a]; }
      const std::vector<double> d = SolveLeastSquares(6, 6, A, b);
      const Mat33 Rn = Mul(Rodrigues({d[0], d[1], d[2]}), R);
      const Vec3 tn = {t[0] + d[3], t[1] + d[4], t[2] + d[5]};
      const double c = cost(Rn, tn);
      if (c < cur) {
        const double gain = cur - c;
        R = Rn; t = tn; cur = c; lambda = std::max(lambda * 0.3, 1e-9); improved = true;
        if (gain < 1e-14 * (cur + 1e-30)) return;
      }
      else lambda *= 10;
    }
    if (!improved) return;
  }
}

// ---- general (non-planar) configuration: four control points -------------------------------------------------
struct Epnp {
  const Problem& p;
  double cw[4][3];
  std::vector<double> alpha;  // [4n]
  double v[4][12];            // null-space candidates, v[0] = smallest eigenvalue
  double L[6][10], rho[6];

  explicit Epnp(const Problem& prob) : p(prob) {}

  void Barycentric()
  {
    Mat33 C;
    for (int r = 0; r < 3; ++r)
      for (int j = 0; j < 3; ++j) C[3 * r + j] = cw[j + 1][r] - cw[0][r];
    const Mat33 Ci = Inverse(C);
    alpha.assign(static_cast<size_t>(4) * p.n, 0.0);
    for (int i = 0; i < p.n; ++i) {
      const Vec3 d = {p.pw[3 * i] - cw[0][0], p.pw[3 * i + 1] - cw[0][1], p.pw[3 * i + 2] - cw[0][2]};
      const Vec3 a = Mul(Ci, d);
      alpha[4 * i + 1] = a[0]; alpha[4 * i + 2] = a[1]; alpha[4 * i + 3] = a[2];
      alpha[4 * i] = 1.0 - a[0] - a[1] - a[2];
    }
  }

  void NullSpace()
  {
    std::vector<double> MtM(144, 0.0);
    for (int i = 0; i < p.n; ++i) {
      double r0[12], r1[12];
      for (int j = 0; j < 4; ++j) {
        const double a = alpha[4 * i + j];
        r0[3 * j] = a; r0[3 * j + 1] = 0; r0[3 * j + 2] = -a * p.uv[2 * i];
        r1[3 * j] = 0; r1[3 * j + 1] = a; r1[3 * j + 2] = -a * p.uv[2 * i + 1];
      }
      for (int a = 0; a < 12; ++a)
        for (int b = 0; b < 12; ++b) MtM[12 * a + b] += r0[a] * r0[b] + r1[a] * r1[b];
    }
    std::vector<double> ev, V;
    EigenSymPSD(12, MtM, ev, V);
    for (int k = 0; k < 4; ++k)
      for (int a = 0; a < 12; ++a) v[k][a] = V[12 * a + (11 - k)];
  }

  void DistanceConstraints()
  {
    static const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
    for (int e = 0; e < 6; ++e) {
      double d[4][3];
      for (int k = 0; k < 4; ++k)
        for (int c = 0; c < 3; ++c) d[k][c] = v[k][3 * pa[e] + c] - v[k][3 * pb[e] + c];
      auto dot = [&](int a, int b) { return d[a][0] * d[b][0] + d[a][1] * d[b][1] + d[a][2] * d[b][2]; };
      // unknowns ordered b11 b12 b22 b13 b23 b33 b14 b24 b34 b44
      L[e][0] = dot(0, 0); L[e][1] = 2 * dot(0, 1); L[e][2] = dot(1, 1); L[e][3] = 2 * dot(0, 2); L[e][4] = 2 * dot(1, 2);
      L[e][5] = dot(2, 2); L[e][6] = 2 * dot(0, 3); L[e][7] = 2 * dot(1, 3); L[e][8] = 2 * dot(2, 3); L[e][9] = dot(3, 3);
      rho[e] = 0;
      for (int c = 0; c < 3; ++c) rho[e] += (cw[pa[e]][c] - cw[pb[e]][c]) * (cw[pa[e]][c] - cw[pb[e]][c]);
    }
  }

  std::vector<double> SolveColumns(const std::vector<int>& cols) const
  {
    const int n = static_cast<int>(cols.size());
    std::vector<double> A(static_cast<size_t>(6) * n), b(rho, rho + 6);
    for (int e = 0; e < 6; ++e)
      for (int j = 0; j < n; ++j) A[static_cast<size_t>(e) * n + j] = L[e][cols[j]];
    return SolveLeastSquares(6, n, A, b);
  }

  // the three linearised initial guesses of the paper (null-space dimension treated as 4-with-one-dominant, 2, 3)
  void InitialBetas(int which, double beta[4]) const
  {
    beta[0] = beta[1] = beta[2] = beta[3] = 0;
    if (which == 0) {
      const std::vector<double> b = SolveColumns({0, 1, 3, 6});
      const double sgn = b[0] < 0 ? -1.0 : 1.0;
      beta[0] = std::sqrt(sgn * b[0]);
      if (beta[0] > 0) { beta[1] = sgn * b[1] / beta[0]; beta[2] = sgn * b[2] / beta[0]; beta[3] = sgn * b[3] / beta[0]; }
      return;
    }
    const std::vector<double> b = which == 1 ? SolveColumns({0, 1, 2}) : SolveColumns({0, 1, 2, 3, 4});
    if (b[0] < 0) { beta[0] = std::sqrt(-b[0]); beta[1] = b[2] < 0 ? std::sqrt(-b[2]) : 0.0; }
    else { beta[0] = std::sqrt(b[0]); beta[1] = b[2] > 0 ? std::sqrt(b[2]) : 0.0; }
    if (b[1] < 0) beta[0] = -beta[0];
    if (which == 2 && beta[0] != 0) beta[2] = b[3] / beta[0];
  }

  void GaussNewton(double beta[4]) const
  {
    for (int it = 0; it < 5; ++it) {
      std::vector<double> J(24), r(6);
      for (int e = 0; e < 6; ++e) {
        const double* l = L[e];
        const double b0 = beta[0], b1 = beta[1], b2 = beta[2], b3 = beta[3];
        J[4 * e + 0] = 2 * l[0] * b0 + l[1] * b1 + l[3] * b2 + l[6] * b3;
        J[4 * e + 1] = l[1] * b0 + 2 * l[2] * b1 + l[4] * b2 + l[7] * b3;
        J[4 * e + 2] = l[3] * b0 + l[4] * b1 + 2 * l[5] * b2 + l[8] * b3;
        J[4 * e + 3] = l[6] * b0 + l[7] * b1 + l[8] * b2 + 2 * l[9] * b3;
        r[e] = rho[e] - (l[0] * b0 * b0 + l[1] * b0 * b1 + l[2] * b1 * b1 + l[3] * b0 * b2 + l[4] * b1 * b2 + l[5] * b2 * b2 +
                         l[6] * b0 * b3 + l[7] * b1 * b3 + l[8] * b2 * b3 + l[9] * b3 * b3);
      }
      const std::vector<double> d = SolveLeastSquares(6, 4, J, r);
      for (int k = 0; k < 4; ++k) beta[k] += d[k];
    }
  }

  double PoseFromBetas(const double beta[4], Mat33& R, Vec3& t) const
  {
    double cc[4][3];
    for (int j = 0; j < 4; ++j)
      for (int c = 0; c < 3; ++c) {
        cc[j][c] = 0;
        for (int k = 0; k < 4; ++k) cc[j][c] += beta[k] * v[k][3 * j + c];
      }
    std::vector<double> pc(static_cast<size_t>(3) * p.n);
    for (int i = 0; i < p.n; ++i)
      for (int c = 0; c < 3; ++c) {
        double s = 0;
        for (int j = 0; j < 4; ++j) s += alpha[4 * i + j] * cc[j][c];
        pc[3 * i + c] = s;
      }
    if (pc[2] < 0)  // the null-space vector is defined up to sign: put the first point in front of the camera
      for (double& x : pc) x = -x;
    AbsoluteOrientation(p, pc, R, t);
    return ReprojError(p, R, t);
  }

  bool Run(const double c0[3], const double axes[9], const double sigma[3], Mat33& R, Vec3& t)
  {
    for (int c = 0; c < 3; ++c) cw[0][c] = c0[c];
    for (int j = 0; j < 3; ++j) {
      const double k = std::sqrt(sigma[j] / p.n);
      for (int c = 0; c < 3; ++c) cw[j + 1][c] = c0[c] + k * axes[3 * c + j];
    }
    Barycentric();
    NullSpace();
    DistanceConstraints();
    double best = std::numeric_limits<double>::infinity();
    for (int which = 0; which < 3; ++which) {
      double beta[4];
      InitialBetas(which, beta);
      GaussNewton(beta);
      Mat33 Rk; Vec3 tk;
      const double err = PoseFromBetas(beta, Rk, tk);
      if (err < best) { best = err; R = Rk; t = tk; }
    }
    return std::isfinite(best);
  }
};

// ---- coplanar configuration: plane -> image homography, then H ~ [r1 r2 t] -----------------------------------
bool PlanarPose(const Problem& p, const double c0[3], const double axes[9], const double sigma[3], Mat33& R, Vec3& t)
{
  const double scale = std::sqrt(sigma[0] / p.n);  // conditions the DLT
  if (!(scale > 0)) return false;
  Vec3 e1 = {axes[0], axes[3], axes[6]}, e2 = {axes[1], axes[4], axes[7]};
  Vec3 e3 = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
  std::vector<double> AtA(81, 0.0), ab(static_cast<size_t>(2) * p.n);
  for (int i = 0; i < p.n; ++i) {
    const double d[3] = {p.pw[3 * i] - c0[0], p.pw[3 * i + 1] - c0[1], p.pw[3 * i + 2] - c0[2]};
    const double a = (d[0] * e1[0] + d[1] * e1[1] + d[2] * e1[2]) / scale, b = (d[0] * e2[0] + d[1] * e2[1] + d[2] * e2[2]) / scale;
    ab[2 * i] = a; ab[2 * i + 1] = b;
    const double u = p.uv[2 * i], w = p.uv[2 * i + 1];
    const double r0[9] = {a, b, 1, 0, 0, 0, -u * a, -u * b, -u};
    const double r1[9] = {0, 0, 0, a, b, 1, -w * a, -w * b, -w};
    for (int x = 0; x < 9; ++x)
      for (int y = 0; y < 9; ++y) AtA[9 * x + y] += r0[x] * r0[y] + r1[x] * r1[y];
  }
  std::vector<double> ev, V;
  EigenSymPSD(9, AtA, ev, V);
  double h[9];
  for (int k = 0; k < 9; ++k) h[k] = V[9 * k + 8];
  // depth of the first point must be positive
  if (h[6] * ab[0] + h[7] * ab[1] + h[8] < 0)
    for (double& x : h) x = -x;
  const double n1 = std::sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]), n2 = std::sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
  if (!(n1 > 0) || !(n2 > 0)) return false;
  const double lambda = 2.0 / (n1 + n2);
  const Vec3 r1 = {lambda * h[0], lambda * h[3], lambda * h[6]}, r2 = {lambda * h[1], lambda * h[4], lambda * h[7]};
  const Vec3 r3 = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
  const Mat33 Rp = NearestRotation({r1[0], r2[0], r3[0], r1[1], r2[1], r3[1], r1[2], r2[2], r3[2]});
  // plane frame -> world: X_plane = E^T (X_w - c0), plane coordinates were divided by `scale`
  const Mat33 Et = {e1[0], e1[1], e1[2], e2[0], e2[1], e2[2], e3[0], e3[1], e3[2]};
  R = Mul(Rp, Et);
  const Vec3 tp = {lambda * h[2] * scale, lambda * h[5] * scale, lambda * h[8] * scale};
  const Vec3 Rc = Mul(R, Vec3{c0[0], c0[1], c0[2]});
  t = {tp[0] - Rc[0], tp[1] - Rc[1], tp[2] - Rc[2]};
  return std::isfinite(ReprojError(p, R, t));
}

}  // namespace

bool SolvePnPEPnP(const std::vector<Point3d>& pts3d, const std::vector<Point2f>& pixels, const Mat33& K, const Vec5& dist,
                  Mat33& R, Vec3& t)
{
  if (pts3d.size() < 4 || pts3d.size() != pixels.size()) return false;
  Problem p;
  p.n = static_cast<int>(pts3d.size());
  const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  // OpenCV's coefficient order read from the reference's vector (see header)
  const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4];
  double c0[3] = {0, 0, 0};
  for (int i = 0; i < p.n; ++i) {
    p.pw.push_back(pts3d[i].x); p.pw.push_back(pts3d[i].y); p.pw.push_back(pts3d[i].z);
    c0[0] += pts3d[i].x; c0[1] += pts3d[i].y; c0[2] += pts3d[i].z;
    const double x0 = (static_cast<double>(pixels[i].x) - cx) / fx, y0 = (static_cast<double>(pixels[i].y) - cy) / fy;
    double x = x0, y = y0;
    for (int it = 0; it < 5; ++it) {
      const double r2 = x * x + y * y;
      const double icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
      if (icdist < 0) { x = x0; y = y0; break; }
      const double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x), dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
      x = (x0 - dx) * icdist;
      y = (y0 - dy) * icdist;
    }
    p.uv.push_back(x); p.uv.push_back(y);
  }
  for (double& c : c0) c /= p.n;
  // principal axes of the point cloud
  std::vector<double> C(9, 0.0);
  for (int i = 0; i < p.n; ++i)
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) C[3 * r + c] += (p.pw[3 * i + r] - c0[r]) * (p.pw[3 * i + c] - c0[c]);
  std::vector<double> sigma, axes;
  EigenSymPSD(3, C, sigma, axes);
  if (!(sigma[1] > 1e-10 * sigma[0])) return false;  // collinear or coincident points
  bool ok;
  if (!(sigma[2] > 1e-8 * sigma[0])) ok = PlanarPose(p, c0, axes.data(), sigma.data(), R, t);
  else {
    Epnp solver(p);
    ok = solver.Run(c0, axes.data(), sigma.data(), R, t);
  }
  if (ok) PolishPose(p, R, t);
  return ok;
}

}  // namespace ptzcalib
