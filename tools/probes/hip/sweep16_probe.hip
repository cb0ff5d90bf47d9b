// Probe (not part of the product): a 16 x 16 Cholesky factorisation inside ONE ROW OF 16 LANES with DPP broadcasts -- lane r holds row r,
// column updates are v_fmac_f64_dpp row_newbcast (no scalar registers, no LDS) -- as the chain-critical part of the diagonal tile's
// factorisation could be done if the rows below were solved on the matrix cores instead of riding along in lanes 16..63.
// Measures cycles per 16-pivot block (one wave, clock64) and checks L L^T = A.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/probes/hip/sweep16_probe tools/probes/hip/sweep16_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

__device__ __forceinline__ double rsqrt_nr(double d)
{
  double y = __builtin_amdgcn_rsq(d);
  y = y * (1.5 - 0.5 * d * y * y);
  y = y * (1.5 - 0.5 * d * y * y);
  return y;
}
__device__ __forceinline__ double rsqrt_cubic(double d)
{
  const double y0 = __builtin_amdgcn_rsq(d);
  const double t = d * y0;
  const double e = fma(-t, y0, 1.0);
  const double p = fma(e, 0.375, 0.5);
  return fma(y0 * e, p, y0);
}
#define BC(dst, src, q) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:" #q " row_mask:0xf bank_mask:0xf" : "=v"(dst) : "v"(src))
#define FD(acc, bsrc, m, q) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #q " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(m))

// (explicit specialisation of a member template of a class template is not allowed: a switch instead)
template <int Q> __device__ __forceinline__ void fd_q(double& acc, const double& l, const double& ml)
{
  if constexpr (Q == 1) FD(acc, l, ml, 1); else if constexpr (Q == 2) FD(acc, l, ml, 2); else if constexpr (Q == 3) FD(acc, l, ml, 3);
  else if constexpr (Q == 4) FD(acc, l, ml, 4); else if constexpr (Q == 5) FD(acc, l, ml, 5); else if constexpr (Q == 6) FD(acc, l, ml, 6);
  else if constexpr (Q == 7) FD(acc, l, ml, 7); else if constexpr (Q == 8) FD(acc, l, ml, 8); else if constexpr (Q == 9) FD(acc, l, ml, 9);
  else if constexpr (Q == 10) FD(acc, l, ml, 10); else if constexpr (Q == 11) FD(acc, l, ml, 11); else if constexpr (Q == 12) FD(acc, l, ml, 12);
  else if constexpr (Q == 13) FD(acc, l, ml, 13); else if constexpr (Q == 14) FD(acc, l, ml, 14); else FD(acc, l, ml, 15);
}
template <int Q> __device__ __forceinline__ void bc_q(double& d, const double& s)
{
  if constexpr (Q == 0) BC(d, s, 0); else if constexpr (Q == 1) BC(d, s, 1); else if constexpr (Q == 2) BC(d, s, 2); else if constexpr (Q == 3) BC(d, s, 3);
  else if constexpr (Q == 4) BC(d, s, 4); else if constexpr (Q == 5) BC(d, s, 5); else if constexpr (Q == 6) BC(d, s, 6); else if constexpr (Q == 7) BC(d, s, 7);
  else if constexpr (Q == 8) BC(d, s, 8); else if constexpr (Q == 9) BC(d, s, 9); else if constexpr (Q == 10) BC(d, s, 10); else if constexpr (Q == 11) BC(d, s, 11);
  else if constexpr (Q == 12) BC(d, s, 12); else if constexpr (Q == 13) BC(d, s, 13); else if constexpr (Q == 14) BC(d, s, 14); else BC(d, s, 15);
}
template <int J, int Q> __device__ __forceinline__ void updates(double (&a)[16], const double& l, const double& ml)
{
  if constexpr (Q < 16) { fd_q<Q>(a[Q], l, ml); updates<J, Q + 1>(a, l, ml); }
}
template <int MODE, int J> __device__ __forceinline__ void sweep(double (&a)[16], double (&ird)[16])
{
  if constexpr (J < 16) {
    double d;
    bc_q<J>(d, a[J]);
    const double r = MODE ? rsqrt_cubic(d) : rsqrt_nr(d);
    ird[J] = r;
    const double l = a[J] * r;
    a[J] = l;
    const double ml = -l;
    asm volatile("s_nop 1");
    updates<J, J + 1>(a, l, ml);
    sweep<MODE, J + 1>(a, ird);
  }
}

// the fused form: every row of 16 lanes carries a REPLICA of the diagonal block's rows (ar: lane r of the row holds block row r) beside
// its own rows of the tile (ap: lane 16 g + r holds tile row 16 g + r); the replicas are factored side by side (same chain, same
// bits), the own rows take the column updates with the replica's broadcasts -- the rows below the diagonal block are solved in the
// same instructions, as in the product's sweep, but without a scalar register on the way
template <int J, int Q> __device__ __forceinline__ void updates2(double (&ar)[16], double (&ap)[16], const double& lr, const double& mlr, const double& mlp)
{
  if constexpr (Q < 16) { fd_q<Q>(ar[Q], lr, mlr); fd_q<Q>(ap[Q], lr, mlp); updates2<J, Q + 1>(ar, ap, lr, mlr, mlp); }
}
template <int MODE, int J> __device__ __forceinline__ void sweep2(double (&ar)[16], double (&ap)[16], double (&ird)[16])
{
  if constexpr (J < 16) {
    double d;
    bc_q<J>(d, ar[J]);
    const double r = MODE ? rsqrt_cubic(d) : rsqrt_nr(d);
    ird[J] = r;
    const double lr = ar[J] * r, lp = ap[J] * r;
    ar[J] = lr; ap[J] = lp;
    const double mlr = -lr, mlp = -lp;
    asm volatile("s_nop 1");
    updates2<J, J + 1>(ar, ap, lr, mlr, mlp);
    sweep2<MODE, J + 1>(ar, ap, ird);
  }
}

// ---- round 6: the fused sweep SOFTWARE-PIPELINED, every instruction of it in inline asm so that the order below IS the issue order:
// pivot J's chain (broadcast of the pivot, rsq + two Newton steps, the scaling of column J) is issued between the column updates of
// pivot J - 1 that it does not depend on (those of the columns behind J), which fill the chain's latency shadows; the negated copies of
// the column are gone (neg modifier of the DPP multiply-add).  Same operations on the same operands: same bits as sweep2.
#define A_RSQ(y, d)        asm volatile("v_rsq_f64 %0, %1" : "=v"(y) : "v"(d))
#define A_MULH(h, d)       asm volatile("v_mul_f64 %0, %1, 0.5" : "=v"(h) : "v"(d))
#define A_MUL(o, a, b)     asm volatile("v_mul_f64 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b))
#define A_FNMA(o, y, t, c) asm volatile("v_fma_f64 %0, -%1, %2, %3" : "=v"(o) : "v"(y), "v"(t), "s"(c))
#define A_MULIP(x, r)      asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(r))
#define FDN(acc, bsrc, m, q) asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:" #q " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(m))
template <int Q> __device__ __forceinline__ void fdn_q(double& acc, const double& l, const double& m)
{
  if constexpr (Q == 1) FDN(acc, l, m, 1); else if constexpr (Q == 2) FDN(acc, l, m, 2); else if constexpr (Q == 3) FDN(acc, l, m, 3);
  else if constexpr (Q == 4) FDN(acc, l, m, 4); else if constexpr (Q == 5) FDN(acc, l, m, 5); else if constexpr (Q == 6) FDN(acc, l, m, 6);
  else if constexpr (Q == 7) FDN(acc, l, m, 7); else if constexpr (Q == 8) FDN(acc, l, m, 8); else if constexpr (Q == 9) FDN(acc, l, m, 9);
  else if constexpr (Q == 10) FDN(acc, l, m, 10); else if constexpr (Q == 11) FDN(acc, l, m, 11); else if constexpr (Q == 12) FDN(acc, l, m, 12);
  else if constexpr (Q == 13) FDN(acc, l, m, 13); else if constexpr (Q == 14) FDN(acc, l, m, 14); else FDN(acc, l, m, 15);
}
// item I of pivot P's update list: column q = P + 1 + I / 2, the replica's row (I even) or the own row (I odd)
template <int P, int I> __device__ __forceinline__ void pf_item(double (&ar)[16], double (&ap)[16])
{
  constexpr int q = P + 1 + I / 2;
  if constexpr (P >= 0 && q < 16) {
    if constexpr ((I & 1) == 0) fdn_q<q>(ar[q], ar[P], ar[P]); else fdn_q<q>(ap[q], ar[P], ap[P]);
  }
}
template <int P, int I0, int I1> __device__ __forceinline__ void pf_fill(double (&ar)[16], double (&ap)[16])
{
  if constexpr (I0 < I1) { pf_item<P, I0>(ar, ap); pf_fill<P, I0 + 1, I1>(ar, ap); }
}
template <int J> __device__ __forceinline__ void sweep2p(double (&ar)[16], double (&ap)[16], double (&ird)[16], const double c15)
{
  if constexpr (J < 16) {
    constexpr int P = J - 1;                      // the pivot whose remaining updates fill this one's chain
    constexpr int NP = P >= 0 ? 2 * (15 - P) : 0; // items 0, 1 (column J) are out already
    double d, y, h, t;
    if constexpr (NP > 2) pf_item<P, 2>(ar, ap); else asm volatile("s_nop 0");   // (a DPP read needs two instructions between it and the write of its source)
    bc_q<J>(d, ar[J]);
    pf_fill<P, 3, 4>(ar, ap);
    A_RSQ(y, d);
    A_MULH(h, d);
    pf_fill<P, 4, 6>(ar, ap);
    A_MUL(t, h, y);
    pf_fill<P, 6, 7>(ar, ap);
    A_FNMA(t, y, t, c15);
    pf_fill<P, 7, 8>(ar, ap);
    A_MUL(y, y, t);
    pf_fill<P, 8, 9>(ar, ap);
    A_MUL(t, h, y);
    pf_fill<P, 9, 10>(ar, ap);
    A_FNMA(t, y, t, c15);
    pf_fill<P, 10, 11>(ar, ap);
    A_MUL(y, y, t);
    ird[J] = y;
    pf_fill<P, 11, 12>(ar, ap);
    A_MULIP(ar[J], y);
    A_MULIP(ap[J], y);
    if constexpr (NP > 12) pf_fill<P, 12, NP>(ar, ap); else asm volatile("s_nop 0");
    pf_fill<J, 0, 2>(ar, ap);                     // column J + 1 is final
    sweep2p<J + 1>(ar, ap, ird, c15);
  }
}
// reference: the product's sweep (lane = row of the 64 x 16 column block, v_readlane broadcasts)
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void sweep_ref(double (&a)[16], double (&ird)[16])
{
  double d = readlane_f64(a[0], 0);
  ird[0] = rsqrt_nr(d);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const double l = a[j] * ird[j];
    a[j] = l;
    if (j + 1 < 16) {
      a[j + 1] -= l * readlane_f64(l, j + 1);
      d = readlane_f64(a[j + 1], j + 1);
      ird[j + 1] = rsqrt_nr(d);
    }
#pragma unroll
    for (int q = j + 2; q < 16; ++q) a[q] -= l * readlane_f64(l, q);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// 64 x 16 column block: rows 0..15 the diagonal block, 16..63 below.  which: 0 reference, 1 fused DPP.
template <int WHICH>
__global__ __launch_bounds__(64) void k_block(const double* A /* [64][16] */, double* L, long long* cyc, int reps)
{
  const int lane = threadIdx.x, r = lane & 15;
  double own0[16], rep0[16];
  for (int q = 0; q < 16; ++q) { own0[q] = (lane < 16 && q > lane) ? 0.0 : A[lane * 16 + q]; rep0[q] = q > r ? 0.0 : A[r * 16 + q]; }
  double ap[16], ar[16], ird[16];
  long long t = 0;
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) { ap[q] = own0[q]; ar[q] = rep0[q]; }
    const long long t0 = clock64();
    if (WHICH == 0) sweep_ref(ap, ird); else if (WHICH == 1) sweep2<0, 0>(ar, ap, ird); else sweep2p<0>(ar, ap, ird, 1.5);
    t += clock64() - t0;
  }
  for (int q = 0; q < 16; ++q) L[lane * 16 + q] = ap[q];
  if (lane == 0) { cyc[0] = t; for (int q = 0; q < 16; ++q) L[64 * 16 + q] = ird[q]; }
}

template <int MODE>
__global__ __launch_bounds__(64) void k_sweep(const double* A, double* L, long long* cyc, int reps)
{
  const int lane = threadIdx.x, r = lane & 15;
  double a0[16];
  for (int q = 0; q < 16; ++q) a0[q] = q > r ? 0.0 : A[r * 16 + q];
  double a[16], ird[16];
  long long t = 0;
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = a0[q];
    const long long t0 = clock64();
    sweep<MODE, 0>(a, ird);
    t += clock64() - t0;
    a0[0] += 1e-13 * a[15];
  }
  if (lane < 16) for (int q = 0; q < 16; ++q) L[r * 16 + q] = a[q];
  if (lane == 0) cyc[0] = t;
}

int main()
{
  std::vector<double> M(16 * 24), A(256);
  unsigned long long s = 777;
  for (auto& v : M) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = ((double)(s >> 11) / 9007199254740992.0) - 0.5; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double t = i == j ? 1e-3 : 0; for (int k = 0; k < 24; ++k) t += M[i * 24 + k] * M[j * 24 + k]; A[i * 16 + j] = t; }
  double *dA, *dL; long long* dc;
  hipMalloc(&dA, 256 * 8); hipMalloc(&dL, 256 * 8); hipMalloc(&dc, 8);
  hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
  const int reps = 1000;
  for (int mode = 0; mode < 2; ++mode) {
    for (int w = 0; w < 2; ++w) {
      if (mode == 0) hipLaunchKernelGGL(k_sweep<0>, dim3(1), dim3(64), 0, 0, dA, dL, dc, reps); else hipLaunchKernelGGL(k_sweep<1>, dim3(1), dim3(64), 0, 0, dA, dL, dc, reps);
      hipDeviceSynchronize();
    }
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    std::vector<double> L(256); hipMemcpy(L.data(), dL, 256 * 8, hipMemcpyDeviceToHost);
    double err = 0, amax = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j <= i; ++j) { double t = 0; for (int k = 0; k <= j; ++k) t += L[i * 16 + k] * L[j * 16 + k]; err = fmax(err, fabs(t - A[i * 16 + j])); amax = fmax(amax, fabs(A[i * 16 + j])); }
    printf("%s: %.0f cycles per 16-pivot block (%.1f per pivot), |LL^T - A| / |A| = %.2e\n", mode ? "cubic rsqrt" : "rsq + 2 Newton", (double)c / reps, (double)c / reps / 16, err / amax);
  }
  {  // the 64 x 16 column block: reference against fused, bit for bit
    std::vector<double> B(64 * 16), L0(64 * 16 + 16), L1(64 * 16 + 16), L2(64 * 16 + 16);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) B[i * 16 + j] = A[i * 16 + j];
    for (int i = 16; i < 64; ++i) for (int j = 0; j < 16; ++j) { s = s * 6364136223846793005ull + 1442695040888963407ull; B[i * 16 + j] = ((double)(s >> 11) / 9007199254740992.0) - 0.5; }
    double *dB, *dLL;
    hipMalloc(&dB, B.size() * 8); hipMalloc(&dLL, L0.size() * 8);
    hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    for (int which = 0; which < 3; ++which) {
      for (int w = 0; w < 2; ++w) {
        if (which == 0) hipLaunchKernelGGL(k_block<0>, dim3(1), dim3(64), 0, 0, dB, dLL, dc, reps);
        else if (which == 1) hipLaunchKernelGGL(k_block<1>, dim3(1), dim3(64), 0, 0, dB, dLL, dc, reps);
        else hipLaunchKernelGGL(k_block<2>, dim3(1), dim3(64), 0, 0, dB, dLL, dc, reps);
        hipDeviceSynchronize();
      }
      long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
      hipMemcpy((which == 2 ? L2 : which ? L1 : L0).data(), dLL, L0.size() * 8, hipMemcpyDeviceToHost);
      printf("64 x 16 column block, %s: %.0f cycles (%.1f per pivot)\n", which == 2 ? "fused DPP sweep, software-pipelined (round 6)" : which ? "fused DPP sweep (replica + own rows)" : "product's sweep (v_readlane)", (double)c / reps, (double)c / reps / 16);
    }
    int diff = 0, diff2 = 0; for (size_t i = 0; i < L0.size(); ++i) { diff += L0[i] != L1[i]; diff2 += L1[i] != L2[i]; }
    printf("values that differ between the first two: %d of %zu; between the fused sweep and its pipelined form: %d\n", diff, L0.size(), diff2);
  }
  return 0;
}
