// ptz_chol.hip -- batched dense Cholesky solve of the reduced camera system (FP64, gfx950).
//
// Replaces, for the PTZ-IBA hot path, the sparse Cholesky that Ceres' SparseSchurComplementSolver
// runs on the Schur complement (linear_solver_type = SPARSE_SCHUR, src/core/ptzray_optimizer.cc:471).
// Any exact factorisation of the damped reduced system yields the same LM step up to round-off; for a
// 360-degree rig the reduced system is ~30 % dense at camera-block level, so it is factored densely.
//
// Layout: right-looking, 64 x 64 tiles, lower triangle, in place in A[count][np][np] (row-major).
//   step k:  chol_panel  : one wave per tile of block column k.  Tile (k,k) is factored in registers
//                          (lane = row, left-looking, L rows broadcast through LDS); every off-diagonal
//                          tile redoes that 64 x 64 factorisation locally (no inter-workgroup wait) and
//                          then solves X L_kk^T = A_ik by substitution.  The factored diagonal tile is
//                          written to Ldiag, never back into A, so concurrent readers see A_kk intact.
//            chol_syrk   : one 256-thread workgroup per trailing tile (i >= j > k):
//                          A_ij -= L_ik L_jk^T with v_mfma_f64_16x16x4_f64, operands staged through LDS.
//   The right-hand side rides along as row n of the padded matrix (diagonal = CHOL_BIG), so the forward
//   substitution is done by the same kernels; chol_backsolve finishes with L^T x = y.
#include "ptz_common.h"

namespace ptz {

namespace {

constexpr int NB = CHOL_NB;
constexpr int LD = NB + 2;  // LDS row stride in doubles: 16-byte aligned rows, conflict-free fragment reads

typedef double d4 __attribute__((ext_vector_type(4)));

// coalesced 64x64 tile copy global (row stride ld) -> LDS (row stride LD); nthreads * 16 B per pass
template <int NTHREADS, bool NEGATE>
__device__ __forceinline__ void tile_g2s(const double* __restrict__ g, int ld, double* s)
{
#pragma unroll
  for (int p = 0; p < (NB * NB / 2) / NTHREADS; ++p) {
    const int idx = p * NTHREADS + threadIdx.x;
    const int row = idx >> 5, c2 = (idx & 31) * 2;
    double2 v = *reinterpret_cast<const double2*>(g + (size_t)row * ld + c2);
    if (NEGATE) { v.x = -v.x; v.y = -v.y; }
    *reinterpret_cast<double2*>(s + row * LD + c2) = v;
  }
}
template <int NTHREADS>
__device__ __forceinline__ void tile_s2g(const double* s, double* __restrict__ g, int ld)
{
#pragma unroll
  for (int p = 0; p < (NB * NB / 2) / NTHREADS; ++p) {
    const int idx = p * NTHREADS + threadIdx.x;
    const int row = idx >> 5, c2 = (idx & 31) * 2;
    *reinterpret_cast<double2*>(g + (size_t)row * ld + c2) = *reinterpret_cast<const double2*>(s + row * LD + c2);
  }
}

// ---- padding: identity rows beyond n, CHOL_BIG at (n, n) --------------------------------------------
__global__ void chol_pad_kernel(CholBatch cb)
{
  const int sys = blockIdx.y;
  if (cb.active && !cb.active[sys]) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = cb.n[sys];
  if (i >= cb.np || i < n) return;
  cb.A[(size_t)sys * cb.np * cb.np + (size_t)i * cb.np + i] = (i == n) ? CHOL_BIG : 1.0;
  if (i == n) cb.fail[sys] = 0;
}

// ---- panel: potrf of the diagonal tile + triangular solve of one off-diagonal tile --------------------
__global__ __launch_bounds__(64) void chol_panel_kernel(CholBatch cb, int k)
{
  const int sys = blockIdx.y;
  if (cb.active && !cb.active[sys]) return;
  const int r = blockIdx.x;  // 0: diagonal tile; r >= 1: tile (k + r, k)
  const int np = cb.np, nt = np / NB;
  const int n = cb.n[sys];
  if (k * NB > n) return;             // whole block column is padding (identity)
  if (r >= 1 && (k + r) * NB > n) return;  // tile rows are all padding (zero below the diagonal)
  double* A = cb.A + (size_t)sys * np * np;
  __shared__ __attribute__((aligned(16))) double Ls[NB * LD];  // L_kk rows (broadcast reads)
  __shared__ __attribute__((aligned(16))) double St[NB * LD];  // staging
  __shared__ double rinv[NB];
  const int lane = threadIdx.x;

  tile_g2s<64, false>(A + (size_t)(k * NB) * np + k * NB, np, St);
  __syncthreads();
  double a[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) a[c] = St[lane * LD + c];
  __syncthreads();

  // left-looking Cholesky, lane = row i:  L[i][j] = (A[i][j] - sum_{k<j} L[i][k] L[j][k]) / L[j][j]
  bool ok = true;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    double s = a[j];
#pragma unroll
    for (int q = 0; q < j; ++q) s -= a[q] * Ls[j * LD + q];
    const double d = __shfl(s, j, WAVE);
    // pivots of padding rows (>= n) are 1 or CHOL_BIG - |y|^2 and never count as failures
    if (!(d > 0.0) && (k * NB + j) < n) ok = false;
    const double ird = 1.0 / sqrt(d);
    const double l = (lane == j) ? d * ird : ((lane > j) ? s * ird : 0.0);
    a[j] = l;
    Ls[lane * LD + j] = l;
    if (lane == j) rinv[j] = ird;
  }
  __syncthreads();

  if (r == 0) {
    // publish the factored diagonal tile (upper part zero) to Ldiag
    double* Ld = cb.Ldiag + ((size_t)sys * nt + k) * (NB * NB);
    tile_s2g<64>(Ls, Ld, NB);
    if (!ok && lane == 0) cb.fail[sys] = 1;
    return;
  }

  // X L_kk^T = A_ik  ->  x[c] = (a_ik[c] - sum_{q<c} x[q] L[c][q]) / L[c][c], lane = row of the tile
  double* T = A + (size_t)((k + r) * NB) * np + k * NB;
  tile_g2s<64, false>(T, np, St);
  __syncthreads();
  double x[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) x[c] = St[lane * LD + c];
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    double s = x[c];
#pragma unroll
    for (int q = 0; q < c; ++q) s -= x[q] * Ls[c * LD + q];
    x[c] = s * rinv[c];
  }
#pragma unroll
  for (int c = 0; c < NB; ++c) St[lane * LD + c] = x[c];
  __syncthreads();
  tile_s2g<64>(St, T, np);
}

// ---- trailing update: A_ij -= L_ik L_jk^T on the matrix cores ---------------------------------------
__global__ __launch_bounds__(256) void chol_syrk_kernel(CholBatch cb, int k)
{
  const int sys = blockIdx.y;
  if (cb.active && !cb.active[sys]) return;
  const int np = cb.np, nt = np / NB;
  const int n = cb.n[sys];
  // linear index -> (i, j), k < j <= i < nt (row-major over the lower triangle of the trailing block)
  const int m = nt - k - 1;
  int t = blockIdx.x;
  int ii = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
  while (ii * (ii + 1) / 2 > t) --ii;
  const int jj = t - ii * (ii + 1) / 2;
  if (ii >= m) return;
  const int ti = k + 1 + ii, tj = k + 1 + jj;
  if (ti * NB > n) return;  // rows of this tile are beyond the rhs row: nothing to update
  double* A = cb.A + (size_t)sys * np * np;
  __shared__ __attribute__((aligned(16))) double As[NB * LD];
  __shared__ __attribute__((aligned(16))) double Bs[NB * LD];
  tile_g2s<256, true>(A + (size_t)(ti * NB) * np + k * NB, np, As);   // -L_ik
  tile_g2s<256, false>(A + (size_t)(tj * NB) * np + k * NB, np, Bs);  //  L_jk
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  // C fragment (f64 16x16x4): element (row = fq + 4 * i, col = fr) of the 16x16 block, i = 0..3
  double* C = A + (size_t)(ti * NB + 16 * w) * np + tj * NB;
  d4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[c][i] = C[(size_t)(fq + 4 * i) * np + 16 * c + fr];
  __syncthreads();
  const double* ap = As + (16 * w + fr) * LD + fq;
  const double* bp = Bs + fr * LD + fq;
#pragma unroll
  for (int kk = 0; kk < NB / 4; ++kk) {
    const double av = ap[4 * kk];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bp[(16 * c) * LD + 4 * kk], acc[c], 0, 0, 0);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) C[(size_t)(fq + 4 * i) * np + 16 * c + fr] = acc[c][i];
}

// ---- back substitution L^T x = y (y = row n of the factored matrix) ---------------------------------
__global__ __launch_bounds__(256) void chol_backsolve_kernel(CholBatch cb, double* xout)
{
  const int sys = blockIdx.y;
  if (cb.active && !cb.active[sys]) return;
  const int np = cb.np, nt = np / NB;
  const int n = cb.n[sys];
  const double* A = cb.A + (size_t)sys * np * np;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* xs = smem;              // [np]
  double* Lt = xs + np;           // [NB * LD]
  double* part = Lt + NB * LD;    // [4 * NB]
  const int tid = threadIdx.x;
  // y = row n of L: its entries in block columns left of the diagonal tile were updated in place, the
  // ones inside the diagonal tile of row n live in Ldiag (diagonal tiles are never written back to A)
  {
    const int kt = n / NB;
    const double* Ldn = cb.Ldiag + ((size_t)sys * nt + kt) * (NB * NB) + (size_t)(n - kt * NB) * NB;
    for (int j = tid; j < np; j += 256) xs[j] = (j < n) ? ((j >= kt * NB) ? Ldn[j - kt * NB] : A[(size_t)n * np + j]) : 0.0;
  }
  __syncthreads();
  for (int k = nt - 1; k >= 0; --k) {
    const int c0 = k * NB;
    if (c0 >= n) continue;
    const int c = tid & 63, g = tid >> 6;
    double p = 0;
    for (int j = c0 + NB + g; j < n; j += 4) p += A[(size_t)j * np + c0 + c] * xs[j];
    part[g * NB + c] = p;
    const double* Ld = cb.Ldiag + ((size_t)sys * nt + k) * (NB * NB);
#pragma unroll
    for (int q = 0; q < (NB * NB / 2) / 256; ++q) {
      const int idx = q * 256 + tid;
      const int row = idx >> 5, c2 = (idx & 31) * 2;
      *reinterpret_cast<double2*>(Lt + row * LD + c2) = *reinterpret_cast<const double2*>(Ld + row * NB + c2);
    }
    __syncthreads();
    if (tid < 64) {
      double rc = xs[c0 + tid] - (part[tid] + part[NB + tid] + part[2 * NB + tid] + part[3 * NB + tid]);
      const double inv = 1.0 / Lt[tid * LD + tid];
      double xc = 0;
      for (int i = NB - 1; i >= 0; --i) {
        double xi = __shfl(rc * inv, i, WAVE);
        if (c0 + i >= n) xi = 0.0;
        if (tid == i) xc = xi;
        if (tid < i) rc -= Lt[i * LD + tid] * xi;
      }
      xs[c0 + tid] = xc;
    }
    __syncthreads();
  }
  for (int j = tid; j < np; j += 256) xout[(size_t)sys * np + j] = (j < n) ? xs[j] : 0.0;
}

}  // namespace

void chol_clear(const CholBatch& cb, hipStream_t stream)
{
  (void)hipMemsetAsync(cb.A, 0, sizeof(double) * (size_t)cb.count * cb.np * cb.np, stream);
  dim3 grid((cb.np + 255) / 256, cb.count);
  hipLaunchKernelGGL(chol_pad_kernel, grid, dim3(256), 0, stream, cb);
}

void chol_panel_launch(const CholBatch& cb, int k, hipStream_t stream)
{
  hipLaunchKernelGGL(chol_panel_kernel, dim3(cb.np / NB - k, cb.count), dim3(64), 0, stream, cb, k);
}
void chol_syrk_launch(const CholBatch& cb, int k, hipStream_t stream)
{
  const int m = cb.np / NB - k - 1;
  if (m > 0) hipLaunchKernelGGL(chol_syrk_kernel, dim3(m * (m + 1) / 2, cb.count), dim3(256), 0, stream, cb, k);
}
void chol_backsolve_launch(const CholBatch& cb, double* x, hipStream_t stream)
{
  const size_t smem = sizeof(double) * ((size_t)cb.np + NB * LD + 4 * NB);
  hipLaunchKernelGGL(chol_backsolve_kernel, dim3(1, cb.count), dim3(256), smem, stream, cb, x);
}

void chol_factor_solve(const CholBatch& cb, double* x, hipStream_t stream)
{
  const int nt = cb.np / NB;
  for (int k = 0; k < nt; ++k) {
    chol_panel_launch(cb, k, stream);
    chol_syrk_launch(cb, k, stream);
  }
  chol_backsolve_launch(cb, x, stream);
}

}  // namespace ptz

// ---- C-ABI test / micro-benchmark entry ---------------------------------------------------------------
extern "C" int32_t ptz_chol_solve_batch(int32_t count, int32_t n, const double* A, const double* rhs, double* x,
                                        int32_t* fail, int32_t device_id, double* device_ms)
{
  using namespace ptz;
  if (count <= 0 || n <= 0 || !A || !rhs || !x) return PTZ_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device_id) return PTZ_ENODEVICE;
  PTZ_HIP_TRY(hipSetDevice(device_id));
  CholBatch cb;
  cb.count = count;
  cb.np = chol_padded_order(n);
  const int np = cb.np, nt = np / CHOL_NB;
  double *dA = nullptr, *dL = nullptr, *dx = nullptr;
  int *dn = nullptr, *dfail = nullptr;
  PTZ_HIP_TRY(hipMalloc(&dA, sizeof(double) * (size_t)count * np * np));
  PTZ_HIP_TRY(hipMalloc(&dL, sizeof(double) * (size_t)count * nt * CHOL_NB * CHOL_NB));
  PTZ_HIP_TRY(hipMalloc(&dx, sizeof(double) * (size_t)count * np));
  PTZ_HIP_TRY(hipMalloc(&dn, sizeof(int) * count));
  PTZ_HIP_TRY(hipMalloc(&dfail, sizeof(int) * count));
  cb.A = dA; cb.Ldiag = dL; cb.n = dn; cb.fail = dfail;
  {
    int* hn = new int[count];
    for (int i = 0; i < count; ++i) hn[i] = n;
    PTZ_HIP_TRY(hipMemcpy(dn, hn, sizeof(int) * count, hipMemcpyHostToDevice));
    delete[] hn;
  }
  hipStream_t stream;
  PTZ_HIP_TRY(hipStreamCreate(&stream));
  hipEvent_t e0, e1;
  PTZ_HIP_TRY(hipEventCreate(&e0));
  PTZ_HIP_TRY(hipEventCreate(&e1));
  chol_clear(cb, stream);
  for (int s = 0; s < count; ++s) {
    PTZ_HIP_TRY(hipMemcpy2DAsync(dA + (size_t)s * np * np, sizeof(double) * np, A + (size_t)s * n * n, sizeof(double) * n,
                                 sizeof(double) * n, n, hipMemcpyHostToDevice, stream));
    PTZ_HIP_TRY(hipMemcpyAsync(dA + (size_t)s * np * np + (size_t)n * np, rhs + (size_t)s * n, sizeof(double) * n,
                               hipMemcpyHostToDevice, stream));
  }
  PTZ_HIP_TRY(hipEventRecord(e0, stream));
  chol_factor_solve(cb, dx, stream);
  PTZ_HIP_TRY(hipEventRecord(e1, stream));
  PTZ_HIP_TRY(hipStreamSynchronize(stream));
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  if (device_ms) *device_ms = ms;
  for (int s = 0; s < count; ++s)
    PTZ_HIP_TRY(hipMemcpy(x + (size_t)s * n, dx + (size_t)s * np, sizeof(double) * n, hipMemcpyDeviceToHost));
  if (fail) PTZ_HIP_TRY(hipMemcpy(fail, dfail, sizeof(int) * count, hipMemcpyDeviceToHost));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(stream);
  (void)hipFree(dA); (void)hipFree(dL); (void)hipFree(dx); (void)hipFree(dn); (void)hipFree(dfail);
  return PTZ_OK;
}
