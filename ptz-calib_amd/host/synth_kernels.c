/* synth_kernels.c -- the inner loop of the synthetic-rig generator (ptz-calib_amd/synth.py make_scene) in C: projection of the
 * candidate (view, ray) pairs and their visibility test.  Plain C, no dependencies; compiled with -ffp-contract=off so that every
 * product and sum is rounded where numpy rounds it: the scenes keep their bits (tests/golden/scene_hashes.json holds both paths).
 * Test / benchmark infrastructure (the data sets of SURVEY 8(d) are synthetic), not part of the calibration path. */
#include <stdint.h>

void ptz_synth_project(int64_t n, const int64_t* vi, const int64_t* pi, const double* Rgt /* [N][9] */, const double* X /* [P][3] */,
                       const double* focal, const double* k1, double cx, double cy, double width, double height,
                       double* u_out, double* v_out, uint8_t* vis_out)
{
  const double ulo = 8.0, uhi = width - 8.0, vlo = 8.0, vhi = height - 8.0;
  for (int64_t i = 0; i < n; ++i) {
    const double* R = Rgt + 9 * vi[i];
    const double* p = X + 3 * pi[i];
    const double X0 = p[0], X1 = p[1], X2 = p[2];
    const double P0 = (R[0] * X0 + R[1] * X1) + R[2] * X2;
    const double P1 = (R[3] * X0 + R[4] * X1) + R[5] * X2;
    const double z = (R[6] * X0 + R[7] * X1) + R[8] * X2;
    const double x = P0 / z, y = P1 / z;
    const double r2 = x * x + y * y;
    const double rad = 1.0 + k1[vi[i]] * r2;
    const double f = focal[vi[i]];
    const double u = (f * x) * rad + cx;
    const double v = (f * y) * rad + cy;
    u_out[i] = u;
    v_out[i] = v;
    vis_out[i] = (uint8_t)((z > 0.1) & (u >= ulo) & (u <= uhi) & (v >= vlo) & (v <= vhi) & (r2 < 1.5));
  }
}

/* (view, ray) candidate pairs of _candidate_pairs: view i's candidates are the rays at the positions [lo[3 i + k], hi[3 i + k]), k = 0..2, of
 * the azimuth order `idx` (the three slices of the numpy form, found there by searchsorted); they are emitted view-major with the ray
 * numbers ASCENDING inside a view -- a bitmap over the rays per view instead of a sort.  offs[i]: where view i's pairs begin. */
#include <string.h>
void ptz_synth_candidates(int64_t P, const int64_t* idx, int64_t N, const int64_t* lo, const int64_t* hi, const int64_t* offs,
                          uint64_t* bitmap /* [(P + 63) / 64] */, int64_t* out_vi, int64_t* out_pi)
{
  const int64_t W = (P + 63) / 64;
  for (int64_t i = 0; i < N; ++i) {
    memset(bitmap, 0, (size_t)W * 8);
    for (int k = 0; k < 3; ++k)
      for (int64_t q = lo[3 * i + k]; q < hi[3 * i + k]; ++q) bitmap[idx[q] >> 6] |= (uint64_t)1 << (idx[q] & 63);
    int64_t o = offs[i];
    for (int64_t w = 0; w < W; ++w) {
      uint64_t m = bitmap[w];
      while (m) {
        const int b = __builtin_ctzll(m);
        out_vi[o] = i; out_pi[o] = (w << 6) + b; ++o;
        m &= m - 1;
      }
    }
  }
}
